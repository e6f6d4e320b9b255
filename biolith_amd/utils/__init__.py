from .fit import FitResult, fit
from .grid_search import GridSearchResult, grid_search_priors
from .init import init_to_feasible, init_to_mean, init_to_median, init_to_sample, init_to_uniform, init_to_value
from .predict import predict

__all__ = ["fit", "FitResult", "predict", "grid_search_priors", "GridSearchResult", "init_to_uniform", "init_to_feasible", "init_to_value",
           "init_to_mean", "init_to_median", "init_to_sample"]
