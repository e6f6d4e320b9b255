from .diagnostics import diagnostics, effective_sample_size, split_gelman_rubin, summary

__all__ = ["diagnostics", "effective_sample_size", "split_gelman_rubin", "summary"]
