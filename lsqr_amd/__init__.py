"""lsqr_amd -- MI355X-native LSQR hot path (lsqr_solver_ez initialize/solve/aprod)."""
__version__ = "0.1.0"
