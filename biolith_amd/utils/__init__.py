from .fit import FitResult, fit
from .predict import predict

__all__ = ["fit", "FitResult", "predict"]
