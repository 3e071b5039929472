from adafortitran_amd.estimators import LinearEstimator  # noqa: F401
