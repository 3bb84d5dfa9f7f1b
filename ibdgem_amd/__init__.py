"""ibdgem_amd -- MI355X-native IBD-likelihood engine (hot path of Paleogenomics/IBDGem).

The product is the C-ABI shared library ibdgem_amd/libibdgem_hip.so
(include/ibdgem_hip.h, sources in ibdgem_amd/csrc) and the C host program in
ibdgem_amd/host.  This Python package is only the ctypes binding used by the
tests and bench.py; it has no compute path of its own and fails loudly when
the library is missing.
"""
from .engine import Engine, EngineError, PinnedArray, load_library, LIB_PATH  # noqa: F401
