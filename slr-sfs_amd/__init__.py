"""slr_sfs_amd -- MI355X-native Euler-warp + softmax-splat frame synthesis.

Drop-in replacement for the hot path of simon3dv/SLR-SFS:
  models/softsplat.py                                  -> slr_sfs_amd.softsplat
  models/projection/euler_integration_manipulator.py   -> slr_sfs_amd.euler_integration_manipulator
  forward_flow data-flow of models/animating_softmax_splating*.py -> slr_sfs_amd.synthesis

All compute runs in hand-written HIP kernels (slr-sfs_amd/csrc, C ABI in include/slr_splat.h)
loaded from slr-sfs_amd/lib/libslrsplat.so.  There is NO fallback: importing works without a
GPU (so the build can be checked), but every operator raises if the library is missing or the
tensors are not on a ROCm device.
"""
from . import _lib  # noqa: F401
from . import softsplat, euler_integration_manipulator, synthesis, nets, pipeline, parallel, io  # noqa: F401
from .softsplat import FunctionSoftsplat, ModuleSoftsplat, ModuleMaximumsplat, ModuleMaximumWarpNormsplat  # noqa: F401
from .euler_integration_manipulator import euler_integration, EulerIntegration, euler_integration_all  # noqa: F401
from .dropin import install_into_reference  # noqa: F401

__version__ = "0.1.0"
