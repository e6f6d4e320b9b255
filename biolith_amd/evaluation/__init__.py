from .diagnostics import diagnostics, effective_sample_size, split_gelman_rubin, summary
from .predictive_checks import deviance, deviance_manual, posterior_predictive_check, residuals
from .predictive_density import log_likelihood, log_likelihood_manual, lppd, lppd_manual, waic, waic_manual

__all__ = ["diagnostics", "effective_sample_size", "split_gelman_rubin", "summary",
           "log_likelihood", "log_likelihood_manual", "lppd", "lppd_manual", "waic", "waic_manual",
           "deviance", "deviance_manual", "posterior_predictive_check", "residuals"]
