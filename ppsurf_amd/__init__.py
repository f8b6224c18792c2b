"""ppsurf_amd -- MI355X-native (gfx950) occupancy-query path of PPSurf behind the reference's Python API.

Hand-written HIP kernels behind a C ABI (include/ppsurf_amd.h, ppsurf_amd/csrc); PyTorch-ROCm is used for device
memory and streams only.  There is no CPU fallback: ops raise `PpsError` when libppsurf_amd.so is missing.
"""
from ._lib import PpsError, LIB_PATH  # noqa: F401

__version__ = '0.1.0'
