"""
ctypes binding of libnegf_hip.so -- exactly the symbols declared in include/negf.h.

There is no CPU fallback: ``load()`` raises if the library has not been built and
``Engine`` (engine.py) raises if no GPU is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NEGF_LIB_PATH: another build of the same library -- A/B runs of kernel variants built with NEGF_EXTRA_HIPCC_FLAGS)
LIB_PATH = os.environ.get("NEGF_LIB_PATH") or os.path.join(_HERE, "lib", "libnegf_hip.so")

NEGF_OK = 0
NEGF_EINVAL = -1
NEGF_ENOMEM = -2
NEGF_EHIP = -3
NEGF_ENODEV = -4
NEGF_ESTATE = -5
NEGF_ESINGULAR = 1
NEGF_IND_TOTAL = -1000
NEGF_SPIN_RESTRICTED = 0
NEGF_SPIN_BLOCK = 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes); must match include/negf.h one to one
SIGNATURES = {
    "negf_device_count": (C.c_int, []),
    "negf_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "negf_destroy": (None, [_vp]),
    "negf_strerror": (C.c_char_p, [C.c_int]),
    "negf_version": (C.c_char_p, []),
    "negf_set_stream": (C.c_int, [_vp, _vp]),
    "negf_set_batch": (C.c_int, [_vp, C.c_int]),
    "negf_get_batch": (C.c_int, [_vp]),
    "negf_set_system": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "negf_set_system_keyed": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_ulonglong]),
    "negf_hash_bytes": (C.c_ulonglong, [_vp, C.c_ulonglong]),
    "negf_sigma_const": (C.c_int, [_vp, C.c_int, _vp, _ip]),
    "negf_sigma_chain1d": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _ip]),
    "negf_sigma_bethe": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _ip]),
    "negf_bethe_raw": (C.c_int, [_vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                 C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "negf_sigma_precomputed": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _ip]),
    "negf_sigma_free": (C.c_int, [_vp, C.c_int]),
    "negf_sigma_eval": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "negf_gr_int": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "negf_gr_int_seg": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "negf_gless_int_seg": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "negf_gr_int_refine": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "negf_gless_int": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "negf_gr_batch": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "negf_transmission": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "negf_dos": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "negf_gr_int_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "negf_gless_int_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "negf_gr_int_seg_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp]),
    "negf_gless_int_seg_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp]),
    "negf_transmission_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "negf_sync": (C.c_int, [_vp]),
    "negf_last_info": (C.c_int, [_vp, C.c_int, _vp]),
    "negf_last_iters": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "negf_set_chain_cache": (C.c_int, [_vp, C.c_int]),
    "negf_set_chain_cache_bytes": (C.c_int, [_vp, C.c_longlong]),
    "negf_chain_cache_clear": (C.c_int, [_vp]),
    "negf_chain_cache_stats": (C.c_int, [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                         C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "negf_profile_enable": (C.c_int, [_vp, C.c_int]),
    "negf_profile_reset": (C.c_int, [_vp]),
    "negf_profile_read": (C.c_int, [_vp, C.c_char_p, _dp, _ip]),
    "negf_profile_read_flops": (C.c_int, [_vp, C.c_char_p, _dp, _dp]),
    "negf_set_inverse_algo": (C.c_int, [_vp, C.c_int]),
    "negf_set_gamma_algo": (C.c_int, [_vp, C.c_int]),
    "negf_set_small_algo": (C.c_int, [_vp, C.c_int]),
    "negf_set_chain_round_robin": (C.c_int, [_vp, C.c_int, C.c_int]),
    "negf_selftest_mfma": (C.c_int, [_vp, _dp]),
}

_lib = None


class NegfError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = load().negf_strerror(code).decode() if _lib is not None else str(code)
        super().__init__(f"libnegf_hip: {where}: {msg} (code {code})")


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(python -m gaunegf_amd.build).  gaunegf_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, where=""):
    """Raise on negative (error) codes; positive codes are numerical conditions
    returned to the caller."""
    if code < 0:
        raise NegfError(code, where)
    return code
