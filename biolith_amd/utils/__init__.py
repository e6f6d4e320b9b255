from .fit import FitResult, fit
from .grid_search import GridSearchResult, grid_search_priors
from .predict import predict

__all__ = ["fit", "FitResult", "predict", "grid_search_priors", "GridSearchResult"]
