"""smcsmc_amd -- MI355X-native particle-filter forward sweep of smcsmc.

The compute path is the HIP library smcsmc_amd/csrc/libsmcsmc_pf.so (C-ABI: include/smcsmc_pf.h);
this package is the thin host-side mirror used by the tests and bench.py.  There is no CPU
fallback: importing works without a GPU, creating a ParticleFilter does not.
"""
from .pf import ParticleFilter, load_library, PfError  # noqa: F401

__version__ = "0.1.0"
