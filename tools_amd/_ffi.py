"""ctypes loader for libpsf_mi355x.so (the C ABI declared in include/psf_mi355x.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is present, every
compute entry point raises.  The oracle under oracle/ is never imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSF_LIB") or os.path.join(_HERE, "lib", "libpsf_mi355x.so")   # PSF_LIB: another build of the same ABI (the experiments build, A/B builds)
EXP_LIB_PATH = os.path.join(_HERE, "lib", "libpsf_mi355x_exp.so")                          # `make exp`: every PSF_* experiment switch alive (tests, tools/); never loaded by default

OK, ERR_PARAM, ERR_NOT_PD, ERR_DOMAIN, ERR_MODULUS, ERR_NO_SOLUTION, ERR_NO_KEY, ERR_HIP, ERR_UNSUPPORTED, ERR_SAMPLER = range(10)


class PsfError(RuntimeError):
    """Non-zero psf_status; the reference panics in the same situations."""

    def __init__(self, status, where=""):
        self.status = status
        msg = lib().psf_status_string(status).decode() if _lib is not None else str(status)
        super().__init__(f"{where}: psf_status {status} ({msg})")


class GadgetParams(C.Structure):
    """psf_gadget_params / GadgetParameters (gadget_parameters.rs:44-52)."""
    _fields_ = [("n", C.c_uint64), ("k", C.c_uint64), ("m_bar", C.c_uint64), ("base", C.c_uint64), ("q", C.c_uint64)]


class PsfpParams(C.Structure):
    _fields_ = [("gp", GadgetParams), ("r", C.c_double), ("s", C.c_double), ("device", C.c_int32), ("flags", C.c_uint32)]


class GpvParams(C.Structure):
    _fields_ = [("gp", GadgetParams), ("s", C.c_double), ("device", C.c_int32), ("flags", C.c_uint32)]


class RingParams(C.Structure):
    _fields_ = [("gp", GadgetParams), ("s", C.c_double), ("s_td", C.c_double), ("device", C.c_int32), ("flags", C.c_uint32)]


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (SONAME
    libamdhip64.so.7, like the system one this library is linked against).  If the system runtime initialises first, torch
    later loads its bundled copy as a second runtime and fails with "No HIP GPUs are available"; if torch's is already in the
    process, the loader satisfies our DT_NEEDED with it and everything shares one runtime (tools/probe_load_order.py shows
    both orders).  So when torch is installed, map its runtime first -- by path, without importing torch.
    PSF_SYSTEM_HIP=1 skips this (hosts that never import torch)."""
    if os.environ.get("PSF_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass                                   # fall back to the system runtime the library was linked against


def open_library(path):
    """CDLL of one build of the library with the few non-default signatures set (the release library, or the experiments build the form-comparison tests load)."""
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). tools_amd has no CPU fallback.")
    _share_torch_hip_runtime()
    L = C.CDLL(path)
    L.psf_status_string.restype = C.c_char_p
    L.psf_status_string.argtypes = [C.c_int]
    L.psfp_m.restype = C.c_size_t
    L.psfp_m.argtypes = [C.c_void_p]
    L.psfp_destroy.restype = None
    L.psfp_destroy.argtypes = [C.c_void_p]
    L.psfgpv_destroy.restype = None
    L.psfgpv_destroy.argtypes = [C.c_void_p]
    L.psfring_destroy.restype = None
    L.psfring_destroy.argtypes = [C.c_void_p]
    return L


def lib():
    global _lib
    if _lib is None:
        _lib = open_library(LIB_PATH)
    return _lib


def check(status, where=""):
    if status != OK:
        raise PsfError(status, where)
