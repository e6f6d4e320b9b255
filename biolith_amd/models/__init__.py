from .occu import OccuSpec, occu, simulate

__all__ = ["occu", "simulate", "OccuSpec"]
