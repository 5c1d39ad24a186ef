"""ctypes front-end of oracle/rans_oracle.c (TEST INFRASTRUCTURE ONLY) and loader for the
compiled reference coder in oracle/_ref (built by oracle/build_oracle.py)."""
import ctypes
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Table(ctypes.Structure):
    _fields_ = [("cdf", ctypes.POINTER(ctypes.c_int32)), ("sizes", ctypes.POINTER(ctypes.c_int32)),
                ("num", ctypes.c_int), ("stride", ctypes.c_int)]


class _Segment(ctypes.Structure):
    _fields_ = [("is_y", ctypes.c_int), ("data", ctypes.c_void_p), ("count", ctypes.c_int),
                ("cdf_offset", ctypes.c_int), ("ch", ctypes.c_int)]


def liboracle():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            from oracle import build_oracle
            build_oracle.build_liboracle()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_rans_encode.restype = ctypes.c_int64
        _LIB.orc_rans_decoder_open.restype = ctypes.c_void_p
    return _LIB


def load_ref():
    """The reference's own MLCodec_extensions_cpp compiled from /root/reference (or None)."""
    path = os.path.join(HERE, "_ref", "MLCodec_extensions_cpp.so")
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("MLCodec_extensions_cpp", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Tables:
    """The two CDF families: index 0 = z (bit estimator), 1 = y (gaussian)."""

    def __init__(self):
        self._keep = [None, None]
        self.arr = (_Table * 2)()

    def set_cdf(self, cdfs, sizes, index):
        cdfs = np.ascontiguousarray(cdfs, dtype=np.int32)
        sizes = np.ascontiguousarray(sizes, dtype=np.int32).reshape(-1)
        self._keep[index] = (cdfs, sizes)
        t = self.arr[index]
        t.cdf = cdfs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        t.sizes = sizes.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        t.num = sizes.size
        t.stride = cdfs.size // sizes.size


def encode(tables, segments, n):
    """segments: list of ('y', int16 array) or ('z', int8 array, cdf_offset, ch), in call order."""
    lib = liboracle()
    segs = (_Segment * len(segments))()
    keep = []
    total = 0
    for i, s in enumerate(segments):
        if s[0] == "y":
            a = np.ascontiguousarray(s[1], dtype=np.int16).reshape(-1)
            segs[i].is_y, segs[i].cdf_offset, segs[i].ch = 1, 0, 1
        else:
            a = np.ascontiguousarray(s[1], dtype=np.int8).reshape(-1)
            segs[i].is_y, segs[i].cdf_offset, segs[i].ch = 0, int(s[2]), int(s[3])
        keep.append(a)
        segs[i].data = a.ctypes.data
        segs[i].count = a.size
        total += a.size
    cap = total * 4 + 16 * 8 + 64
    out = np.zeros(cap, dtype=np.uint8)
    size = lib.orc_rans_encode(tables.arr, segs, len(segments), int(n),
                               out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(cap))
    assert 0 <= size <= cap
    return out[:size].copy()


class Decoder:
    def __init__(self, tables, stream, n):
        self.lib = liboracle()
        self.tables = tables
        self._stream = np.ascontiguousarray(stream, dtype=np.uint8)
        self.h = ctypes.c_void_p(self.lib.orc_rans_decoder_open(
            self._stream.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(self._stream.size), int(n)))

    def decode_y(self, indexes):
        idx = np.ascontiguousarray(indexes, dtype=np.uint8).reshape(-1)
        out = np.empty(idx.size, dtype=np.int8)
        self.lib.orc_rans_decode_y(self.h, self.tables.arr, idx.ctypes.data_as(ctypes.c_void_p),
                                   idx.size, out.ctypes.data_as(ctypes.c_void_p))
        return out

    def decode_z(self, count, cdf_offset, ch):
        out = np.empty(int(count), dtype=np.int8)
        self.lib.orc_rans_decode_z(self.h, self.tables.arr, int(count), int(cdf_offset), int(ch),
                                   out.ctypes.data_as(ctypes.c_void_p))
        return out

    def close(self):
        if self.h:
            self.lib.orc_rans_decoder_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


def pmf_to_quantized_cdf(pmf):
    p = np.ascontiguousarray(pmf, dtype=np.float32).reshape(-1)
    out = np.zeros(p.size + 1, dtype=np.uint32)
    rc = liboracle().orc_pmf_to_quantized_cdf(p.ctypes.data_as(ctypes.c_void_p), p.size,
                                               out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return [int(v) for v in out]
