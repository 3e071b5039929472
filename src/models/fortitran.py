from adafortitran_amd.estimators import BaseFortiTranEstimator, FortiTranEstimator  # noqa: F401
