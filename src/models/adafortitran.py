from adafortitran_amd.estimators import AdaFortiTranEstimator  # noqa: F401
