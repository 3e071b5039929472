from adafortitran_amd.estimators import AdaFortiTranEstimator, FortiTranEstimator, LinearEstimator  # noqa: F401
