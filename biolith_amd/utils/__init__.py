from .fit import FitResult, fit

__all__ = ["fit", "FitResult"]
