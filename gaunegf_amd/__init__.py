"""
gaunegf_amd -- MI355X-native NEGF energy-grid engine behind the GauNEGF Python API.

Module names follow the reference (``gauNEGF.integrate`` -> ``gaunegf_amd.integrate``
...), so ``from gauNEGF.integrate import GrInt`` becomes
``from gaunegf_amd.integrate import GrInt``.  All O(N^3) per-energy work runs in
hand-written HIP kernels (libnegf_hip.so, C ABI in include/negf.h); Python keeps the
grid / weight / checkpoint bookkeeping.  There is no CPU fallback.
"""
from . import config  # noqa: F401

__version__ = "0.1.0"
