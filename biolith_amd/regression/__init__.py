"""Regressor tokens (biolith/regression/__init__.py).  Only ``LinearRegression`` is built:
``intercept + covs @ slopes`` (biolith/regression/linear.py:48-66) is inlined in the HIP kernel."""


class AbstractRegression:  # biolith/regression/abstract.py:8-20
    def __init__(self, name, n_covs, prior=None):
        self.name, self.n_covs, self.prior = name, n_covs, prior

    def __call__(self, covs):
        raise NotImplementedError


class LinearRegression(AbstractRegression):
    """Marker for the linear predictor evaluated inside the occupancy kernel."""


__all__ = ["AbstractRegression", "LinearRegression"]
