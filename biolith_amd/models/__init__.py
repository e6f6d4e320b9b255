from .nmixture import nmixture, simulate_nmixture
from .occu import OccuSpec, occu, simulate
from .occu_cop import occu_cop, simulate_cop
from .occu_cs import occu_cs, simulate_cs
from .occu_dyn import occu_dyn, simulate_dyn
from .occu_rn import occu_rn, simulate_rn

__all__ = ["occu", "simulate", "occu_rn", "simulate_rn", "occu_cop", "simulate_cop", "nmixture", "simulate_nmixture",
           "occu_cs", "simulate_cs", "occu_dyn", "simulate_dyn", "OccuSpec"]
