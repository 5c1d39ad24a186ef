"""Drop-in for the reference pybind11 module ``MLCodec_extensions_cpp``
(/root/reference/src/cpp/py_rans/bind.cpp:14-40): ``RansEncoder``, ``RansDecoder`` and
``pmf_to_quantized_cdf`` with the same method names and argument meaning, implemented over the
C ABI of libdcvc_amd.so (include/dcvc_amd_rans.h).

Put ``dcvc_amd/plugin`` on ``sys.path`` (or call ``dcvc_amd.install_plugin()``) and the
reference's ``src/models/entropy_models.py:30-43`` imports this module unchanged.
"""
import ctypes

import numpy as np

from dcvc_amd import _lib

_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_c_i8p = ctypes.POINTER(ctypes.c_int8)
_c_i16p = ctypes.POINTER(ctypes.c_int16)
_c_u32p = ctypes.POINTER(ctypes.c_uint32)
_c_f32p = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p


def _sig():
    f = _lib.fn
    return dict(
        pmf=f("dcvc_pmf_to_quantized_cdf", ctypes.c_int, [_c_f32p, ctypes.c_int, _c_u32p]),
        e_new=f("dcvc_rans_encoder_create", _vp, []),
        e_del=f("dcvc_rans_encoder_destroy", None, [_vp]),
        e_cdf=f("dcvc_rans_encoder_set_cdf", ctypes.c_int,
                [_vp, _c_i32p, ctypes.c_int, ctypes.c_int, _c_i32p, ctypes.c_int]),
        e_par=f("dcvc_rans_encoder_set_entropy_coder_parallel", ctypes.c_int, [_vp, ctypes.c_int]),
        e_reset=f("dcvc_rans_encoder_reset", ctypes.c_int, [_vp]),
        e_y=f("dcvc_rans_encoder_encode_y", ctypes.c_int, [_vp, _c_i16p, ctypes.c_int]),
        e_z=f("dcvc_rans_encoder_encode_z", ctypes.c_int,
              [_vp, _c_i8p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
        e_flush=f("dcvc_rans_encoder_flush", ctypes.c_int, [_vp]),
        e_get=f("dcvc_rans_encoder_get_encoded_stream", ctypes.c_int64,
                [_vp, _c_u8p, ctypes.c_size_t]),
        d_new=f("dcvc_rans_decoder_create", _vp, []),
        d_del=f("dcvc_rans_decoder_destroy", None, [_vp]),
        d_cdf=f("dcvc_rans_decoder_set_cdf", ctypes.c_int,
                [_vp, _c_i32p, ctypes.c_int, ctypes.c_int, _c_i32p, ctypes.c_int]),
        d_par=f("dcvc_rans_decoder_set_entropy_coder_parallel", ctypes.c_int, [_vp, ctypes.c_int]),
        d_stream=f("dcvc_rans_decoder_set_stream", ctypes.c_int, [_vp, _c_u8p, ctypes.c_size_t]),
        d_y=f("dcvc_rans_decoder_decode_y", ctypes.c_int, [_vp, _c_u8p, ctypes.c_int, _c_i8p]),
        d_z=f("dcvc_rans_decoder_decode_z", ctypes.c_int,
              [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i8p]),
    )


_F = _sig()


def _arr(a, dtype):
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


def _cdf_args(cdfs, cdfs_sizes):
    cdfs = _arr(cdfs, np.int32)
    sizes = _arr(cdfs_sizes, np.int32).reshape(-1)
    num = sizes.size
    stride = cdfs.size // num
    return cdfs, sizes, num, stride


def pmf_to_quantized_cdf(pmf):
    """list[float] -> list[int] of len(pmf) + 1 (bind.cpp:40)."""
    p = _arr(pmf, np.float32).reshape(-1)
    out = np.zeros(p.size + 1, dtype=np.uint32)
    _lib.check(_F["pmf"](p.ctypes.data_as(_c_f32p), p.size, out.ctypes.data_as(_c_u32p)))
    return [int(v) for v in out]


class RansEncoder:
    def __init__(self):
        self._h = _F["e_new"]()
        if not self._h:
            raise _lib.DcvcError("cannot create rANS encoder")

    def __del__(self, _destroy=_F["e_del"]):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _destroy(h)

    def set_cdf(self, cdfs, cdfs_sizes, index):
        cdfs, sizes, num, stride = _cdf_args(cdfs, cdfs_sizes)
        _lib.check(_F["e_cdf"](self._h, cdfs.ctypes.data_as(_c_i32p), num, stride,
                               sizes.ctypes.data_as(_c_i32p), int(index)))

    def set_entropy_coder_parallel(self, n):
        _lib.check(_F["e_par"](self._h, int(n)))

    def reset(self):
        _lib.check(_F["e_reset"](self._h))

    def encode_y(self, symbols):
        s = _arr(symbols, np.int16).reshape(-1)
        _lib.check(_F["e_y"](self._h, s.ctypes.data_as(_c_i16p), s.size))

    def encode_z(self, symbols, cdf_offset, ch):
        s = _arr(symbols, np.int8).reshape(-1)
        _lib.check(_F["e_z"](self._h, s.ctypes.data_as(_c_i8p), s.size, int(cdf_offset), int(ch)))

    def flush(self):
        _lib.check(_F["e_flush"](self._h))

    def get_encoded_stream(self):
        n = _F["e_get"](self._h, None, 0)
        out = np.empty(n, dtype=np.uint8)
        _F["e_get"](self._h, out.ctypes.data_as(_c_u8p), n)
        return out


class RansDecoder:
    def __init__(self):
        self._h = _F["d_new"]()
        if not self._h:
            raise _lib.DcvcError("cannot create rANS decoder")
        self._decoded = np.zeros(0, dtype=np.int8)

    def __del__(self, _destroy=_F["d_del"]):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _destroy(h)

    def set_cdf(self, cdfs, cdfs_sizes, index):
        cdfs, sizes, num, stride = _cdf_args(cdfs, cdfs_sizes)
        _lib.check(_F["d_cdf"](self._h, cdfs.ctypes.data_as(_c_i32p), num, stride,
                               sizes.ctypes.data_as(_c_i32p), int(index)))

    def set_entropy_coder_parallel(self, n):
        _lib.check(_F["d_par"](self._h, int(n)))

    def set_stream(self, encoded):
        s = _arr(encoded, np.uint8).reshape(-1)
        _lib.check(_F["d_stream"](self._h, s.ctypes.data_as(_c_u8p), s.size))

    def decode_y(self, indexes):
        idx = _arr(indexes, np.uint8).reshape(-1)
        self._decoded = np.empty(idx.size, dtype=np.int8)
        _lib.check(_F["d_y"](self._h, idx.ctypes.data_as(_c_u8p), idx.size,
                             self._decoded.ctypes.data_as(_c_i8p)))

    def decode_z(self, total_size, cdf_offset, ch):
        self._decoded = np.empty(int(total_size), dtype=np.int8)
        _lib.check(_F["d_z"](self._h, int(total_size), int(cdf_offset), int(ch),
                             self._decoded.ctypes.data_as(_c_i8p)))

    def get_decoded_tensor(self):
        """int8 result of the last decode_y / decode_z (the reference only exposes this to C++,
        py_rans.h:60 ``get_decoded_tensor_cpp``)."""
        return self._decoded
