"""biolith_amd -- MI355X-native occupancy-model NUTS engine behind biolith's fit()/occu/simulate API.

    from biolith_amd.models import occu, simulate
    from biolith_amd.utils import fit
    data, truth = simulate()
    result = fit(occu, **data)

Only the hot path ``fit(occu, ...)`` of timmh/biolith is built: host code
here is a thin mirror of the reference interface; the log-density, its gradient, the NUTS sampler
and the RNG run in hand-written gfx950 kernels behind ``include/biolith_hip.h``.
"""
__version__ = "0.1.0"
