"""ctypes loader for libdcvc_amd.so (the C-ABI boundary, include/*.h).

The library is built in-tree by ``python -m dcvc_amd.build`` (or ``__graft_entry__.build()``).
There is no fallback: if the shared object is missing or a symbol cannot be resolved the import
fails loudly - the product never routes through a CPU restatement.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# DCVC_LIB selects another build of the same library (tuning variants, dcvc_amd/build.py)
LIB_PATH = os.environ.get("DCVC_LIB") or os.path.join(_PKG, "libdcvc_amd.so")

_lib = None


class DcvcError(RuntimeError):
    """Raised when a libdcvc_amd entry point reports an error (mirrors the reference's C++
    exception -> RuntimeError translation, cuda_check.h:10-17)."""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libdcvc_amd.so is not built (%s). Run `python -m dcvc_amd.build` - there is no "
                "CPU fallback for the codec path." % LIB_PATH)
        # torch must own the HIP runtime of this process (same soname, libamdhip64.so.7): import
        # it first when it is installed so both sides share one runtime instance.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the pure host coder
            pass
        _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _lib.dcvc_last_error.restype = ctypes.c_char_p
    return _lib


def check(rc):
    if rc is None:
        return
    if isinstance(rc, int) and rc < 0:
        raise DcvcError(lib().dcvc_last_error().decode("utf-8", "replace"))
    return rc


def fn(name, restype, argtypes):
    """Resolve one exported symbol with its signature; raises AttributeError if absent."""
    f = getattr(lib(), name)
    f.restype = restype
    f.argtypes = argtypes
    return f
