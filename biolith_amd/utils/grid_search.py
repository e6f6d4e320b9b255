"""``grid_search_priors`` -- drop-in for biolith/utils/grid_search.py:116-516 on the HIP engine.

The reference tries Normal and Laplace priors over a grid of (loc, scale) for the occupancy and the detection
coefficients, scores each combination by the mean validation LPPD over a stratified k-fold split of the sites (sites with /
without a detection), and refits the best one on all data.  Each fold there runs in a spawned process to keep XLA's memory in
check and to enforce the timeout; here a fold is three in-process calls -- ``fit`` on the training sites, ``predict`` on the
validation sites, ``lppd`` -- and the timeout is the one ``fit`` / ``predict`` already honour (utils/misc.py).
"""
from __future__ import annotations

import itertools
import warnings
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Union

import numpy as np

from ..distributions import Laplace, Normal
from ..evaluation import lppd
from .fit import FitResult, fit
from .predict import predict


class GridSearchResult(NamedTuple):
    best_result: FitResult
    best_params: Dict[str, Any]
    best_score: float
    cv_results: List[Dict[str, Any]]


_FAMILIES = {"normal": Normal, "laplace": Laplace}
_DEFAULT_GRID = {"loc": [0.0], "scale": [0.25, 0.5, 1.0, 2.0, 4.0]}
_SINGLE = {"loc": [0.0], "scale": [1.0]}


def grid_search_priors(
    model_fn: Callable,
    site_covs,
    obs_covs,
    obs,
    regressor_occ: Any,
    regressor_det: Any,
    prior_types: Optional[List[str]] = None,
    prior_params_occ: Union[Dict[str, Dict[str, List[float]]], bool, None] = None,
    prior_params_det: Union[Dict[str, Dict[str, List[float]]], bool, None] = None,
    cv_folds: int = 5,
    random_seed: int = 42,
    num_samples: int = 1000,
    num_warmup: int = 1000,
    num_chains: int = 5,
    kernel: Optional[str] = None,
    init_strategy: Optional[Callable] = None,
    timeout: Optional[int] = None,
    **kwargs,
) -> GridSearchResult:
    """Grid search over coefficient priors by stratified k-fold cross-validation on the validation LPPD.

    Parameters, defaults and the returned :class:`GridSearchResult` are those of the reference
    (biolith/utils/grid_search.py:116-290): ``prior_types`` defaults to ``["normal", "laplace"]``; ``prior_params_occ`` /
    ``prior_params_det`` are ``{family: {"loc": [...], "scale": [...]}}`` (``None``: loc 0, scales 0.25 … 4; ``False``: that side
    stays at ``(0, 1)``); folds are stratified by "site has at least one detection"; every (family, occupancy parameters,
    detection parameters) combination is scored by its mean validation LPPD and the best one is refitted on all sites.
    ``coords`` / ``ell`` in ``**kwargs`` (what ``simulate()`` returns next to the data) are passed through to the model.
    """
    try:
        from sklearn.model_selection import StratifiedKFold
    except ImportError as e:   # grid_search.py:311-316
        raise ImportError("sklearn is required for grid search. Please install before using grid search, "
                          "e.g. using 'pip install scikit-learn'.") from e

    if prior_types is None:
        prior_types = ["normal", "laplace"]
    for ptype in prior_types:
        if ptype not in _FAMILIES:
            raise ValueError(f"Unsupported prior type: {ptype}. Must be one of {list(_FAMILIES)}.")
    if prior_params_occ is None:
        prior_params_occ = {t: dict(_DEFAULT_GRID) for t in _FAMILIES}
    elif prior_params_occ is False:
        prior_params_occ = {t: dict(_SINGLE) for t in prior_types}
    if prior_params_det is None:
        prior_params_det = dict(prior_params_occ)
    elif prior_params_det is False:
        prior_params_det = {t: dict(_SINGLE) for t in prior_types}

    site_covs, obs_covs, obs = np.asarray(site_covs), np.asarray(obs_covs), np.asarray(obs)
    stratify = (np.nansum(obs, axis=(0, 2, 3)) > 0).astype(int)
    if len(np.unique(stratify)) == 1:
        warnings.warn(f"All sites have the same occupancy status ({stratify[0]}). Stratification will not be effective.")
    cv = StratifiedKFold(n_splits=cv_folds, shuffle=True, random_state=random_seed)
    folds = list(cv.split(np.arange(site_covs.shape[0]), stratify))
    common = dict(regressor_occ=regressor_occ, regressor_det=regressor_det, **kwargs)
    # kernel / init_strategy belong to fit() alone (grid_search.py:64-96): predict() and lppd() hand unknown keywords to the model
    fit_only = {}
    if kernel is not None:
        fit_only["kernel"] = kernel
    if init_strategy is not None:
        fit_only["init_strategy"] = init_strategy

    best_score, best_params, best_priors, cv_results = float("-inf"), {}, None, []
    for prior_type in prior_types:
        occ_grid, det_grid = prior_params_occ.get(prior_type, {}), prior_params_det.get(prior_type, {})
        if not occ_grid and not det_grid:
            warnings.warn(f"No parameters found for prior type '{prior_type}'. Skipping.")
            continue
        occ_grid, det_grid = occ_grid or dict(_SINGLE), det_grid or dict(_SINGLE)
        family = _FAMILIES[prior_type]
        for occ_vals, det_vals in itertools.product(itertools.product(*occ_grid.values()), itertools.product(*det_grid.values())):
            occ_params, det_params = dict(zip(occ_grid, occ_vals)), dict(zip(det_grid, det_vals))
            prior_occ, prior_det = family(occ_params["loc"], occ_params["scale"]), family(det_params["loc"], det_params["scale"])
            fold_scores = []
            for fold_idx, (train, val) in enumerate(folds):
                try:
                    trained = fit(model_fn, site_covs=site_covs[train], obs_covs=obs_covs[train], obs=obs[:, train],
                                  prior_beta=prior_occ, prior_alpha=prior_det, num_samples=num_samples, num_warmup=num_warmup,
                                  num_chains=num_chains, random_seed=random_seed + fold_idx, timeout=timeout, **fit_only, **common)
                    held_out = dict(site_covs=site_covs[val], obs_covs=obs_covs[val], obs=obs[:, val],
                                    prior_beta=prior_occ, prior_alpha=prior_det, **common)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore", UserWarning)   # Predictive's num_samples note
                        predictions = predict(model_fn, trained.mcmc, timeout=timeout, **held_out)
                    score = float(lppd(model_fn, predictions, **held_out))
                    if np.isfinite(score):
                        fold_scores.append(score)
                    else:
                        warnings.warn(f"Invalid LPPD score ({score}) in fold {fold_idx}")
                except Exception as e:  # grid_search.py:443-445: a failed fold is skipped with a warning
                    warnings.warn(f"Model fit failed in fold {fold_idx}: {e}")
            if not fold_scores:
                warnings.warn(f"No successful folds for parameters: prior_type={prior_type}, occ={occ_params}, det={det_params}")
                continue
            mean_score = float(np.mean(fold_scores))
            cv_results.append(dict(prior_type=prior_type, occ_params=occ_params, det_params=det_params, mean_val_lppd=mean_score,
                                   std_val_lppd=float(np.std(fold_scores)), fold_scores=fold_scores,
                                   n_successful_folds=len(fold_scores)))
            if mean_score > best_score:
                best_score = mean_score
                best_params = dict(prior_type=prior_type, occ_params=occ_params, det_params=det_params)
                best_priors = (prior_occ, prior_det)
    if best_priors is None:
        raise RuntimeError("Grid search failed: no successful parameter combinations found.")
    # the reference refits every time the best score improves and keeps the last refit (grid_search.py:471-504): same result,
    # one fit
    best_result = fit(model_fn, site_covs=site_covs, obs_covs=obs_covs, obs=obs, prior_beta=best_priors[0], prior_alpha=best_priors[1],
                      num_samples=num_samples, num_warmup=num_warmup, num_chains=num_chains, random_seed=random_seed, timeout=timeout,
                      **fit_only, **common)
    return GridSearchResult(best_result, best_params, best_score, cv_results)
