"""thallo_amd -- MI355X-native Gauss-Newton / LM + PCG backend behind the Thallo C API.

Layout: csrc/ (gfx950 HIP kernels, C-ABI shim, C++ driver, Thallo.h entry points),
energies/ (bundled .t problem specifications), api.py (ctypes mirror of Thallo.h),
synthetic.py (seeded problem instances).  The compute path is libThallo.so only.
"""
from .api import ThalloSolver, lib, energy_file, last_error  # noqa: F401
