from .occu import OccuSpec, occu, simulate
from .occu_rn import occu_rn, simulate_rn

__all__ = ["occu", "simulate", "occu_rn", "simulate_rn", "OccuSpec"]
