"""`lib.pnp` call surface: `cer_solver.solve` (torch-facing) and `pnp_ceres.solve` (marshaller)."""
from . import cer_solver, pnp_ceres, gpu_solver  # noqa: F401
