"""Native bit-stream container (include/dcvc_amd_stream.h) against the Python
mirror of stream_helper.py (dcvc_amd/stream_helper.py, itself checked against bytes produced by the
reference's own module in tests/test_io_cpu.py) and, when /root/reference is present, against the
reference module directly. CPU only: the library loads and the host code runs without a GPU."""
import ctypes
import importlib.util
import io
import os

import numpy as np
import pytest

from dcvc_amd import _lib, stream_helper as sh

u8p = ctypes.POINTER(ctypes.c_uint8)
ci, cz = ctypes.c_int, ctypes.c_size_t


@pytest.fixture(scope="module")
def nat():
    f = lambda name, res, args: _lib.fn(name, res, args)
    return dict(
        write_uint=f("dcvc_stream_write_uint", ci, [ctypes.c_void_p, cz, ctypes.c_uint32]),
        read_uint=f("dcvc_stream_read_uint", ci, [ctypes.c_void_p, cz, ctypes.POINTER(ctypes.c_uint32)]),
        write_sps=f("dcvc_stream_write_sps", ci, [ctypes.c_void_p, cz, ci, ci, ci]),
        write_ip=f("dcvc_stream_write_ip", ctypes.c_int64, [ctypes.c_void_p, cz, ci, ci, ci, ci, ci, ctypes.c_void_p, cz]),
        read_header=f("dcvc_stream_read_header", ci, [ctypes.c_void_p, cz, ctypes.POINTER(ci), ctypes.POINTER(ci)]),
        read_sps=f("dcvc_stream_read_sps_remaining", ci, [ctypes.c_void_p, cz, ctypes.POINTER(ci), ctypes.POINTER(ci)]),
        read_ip=f("dcvc_stream_read_ip_remaining", ctypes.c_int64,
                  [ctypes.c_void_p, cz, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci),
                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(cz)]),
    )


def _buf(n):
    return (ctypes.c_uint8 * n)()


def _sequence():
    rng = np.random.default_rng(5)
    units = [("sps", 0, 1080, 1920)]
    for i in range(7):
        n = int(rng.choice([0, 1, 100, 127, 128, 5000, 16383, 16384, 70000]))
        units.append(("ip", i == 0, 0, int(rng.integers(0, 64)), int(rng.integers(1, 9)), int(rng.integers(0, 2)),
                      rng.integers(0, 256, n, dtype=np.uint8).tobytes()))
    units.insert(4, ("sps", 1, 2160, 3840))
    units.append(("ip", False, 1, 63, 8, 1, b"\x01\x02\x03"))
    return units


def _python_bytes(units, mod=sh):
    f = io.BytesIO()
    for u in units:
        if u[0] == "sps":
            mod.write_sps(f, {"sps_id": u[1], "height": u[2], "width": u[3]})
        else:
            mod.write_ip(f, u[1], u[2], u[3], u[4], u[5], u[6])
    return f.getvalue()


def _native_bytes(nat, units):
    out = b""
    for u in units:
        if u[0] == "sps":
            b = _buf(16)
            n = nat["write_sps"](b, 16, u[1], u[2], u[3])
            assert n > 0
        else:
            size = nat["write_ip"](None, 0, int(u[1]), u[2], u[3], u[4], u[5], u[6], len(u[6]))
            b = _buf(size)
            n = nat["write_ip"](b, size, int(u[1]), u[2], u[3], u[4], u[5], u[6], len(u[6]))
            assert n == size
        out += bytes(b[:n])
    return out


def test_varuint_boundaries(nat):
    for v in (0, 1, 127, 128, 16383, 16384, 2 ** 20 + 3, 2 ** 30 - 1):
        b = _buf(8)
        n = nat["write_uint"](b, 8, v)
        f = io.BytesIO()
        assert sh.write_uint_adaptive(f, v) == n and bytes(b[:n]) == f.getvalue()
        got = ctypes.c_uint32()
        assert nat["read_uint"](b, n, ctypes.byref(got)) == n and got.value == v
    assert nat["write_uint"](_buf(8), 8, 2 ** 30) == -1              # does not fit 30 bits
    assert nat["write_uint"](_buf(1), 1, 300) == -1                  # destination too small
    got = ctypes.c_uint32()
    assert nat["read_uint"](bytes([0xC1, 0x00]), 2, ctypes.byref(got)) == -2     # truncated


def test_writer_matches_python_mirror_byte_for_byte(nat):
    units = _sequence()
    assert _native_bytes(nat, units) == _python_bytes(units)


def test_reader_parses_python_written_stream(nat):
    units = _sequence()
    data = _python_bytes(units)
    pos, seen = 0, []
    while pos < len(data):
        nal, sid = ci(), ci()
        k = nat["read_header"](data[pos:], len(data) - pos, ctypes.byref(nal), ctypes.byref(sid))
        assert k == 1
        pos += k
        if nal.value == 0:
            h, w = ci(), ci()
            k = nat["read_sps"](data[pos:], len(data) - pos, ctypes.byref(h), ctypes.byref(w))
            seen.append(("sps", sid.value, h.value, w.value))
        else:
            qp, ec, rs, n = ci(), ci(), ci(), cz()
            chunk = (ctypes.c_uint8 * (len(data) - pos)).from_buffer_copy(data[pos:])
            pl = ctypes.c_void_p()
            k = nat["read_ip"](chunk, len(chunk), ctypes.byref(qp), ctypes.byref(ec), ctypes.byref(rs),
                               ctypes.byref(pl), ctypes.byref(n))
            payload = ctypes.string_at(pl.value, n.value)
            assert pl.value + n.value == ctypes.addressof(chunk) + k          # the payload ends the unit
            seen.append(("ip", nal.value == 1, sid.value, qp.value, ec.value, rs.value, payload))
        assert k > 0
        pos += k
    assert seen == [tuple(u) for u in units]


def test_errors(nat):
    nal, sid = ci(), ci()
    assert nat["read_header"](b"", 0, ctypes.byref(nal), ctypes.byref(sid)) == -2
    assert nat["read_header"](bytes([0x70]), 1, ctypes.byref(nal), ctypes.byref(sid)) == -3       # nal_type 7
    assert nat["write_sps"](_buf(16), 16, 16, 64, 64) == -1                                      # sps_id > 15
    assert nat["write_ip"](_buf(16), 16, 1, 0, 300, 1, 0, b"", 0) == -1                          # qp > 255
    qp, ec, rs, n, pl = ci(), ci(), ci(), cz(), ctypes.c_void_p()
    short = bytes([32, 3, 10, 1, 2, 3])            # announces 10 payload bytes, carries 3
    assert nat["read_ip"](short, len(short), ctypes.byref(qp), ctypes.byref(ec), ctypes.byref(rs),
                          ctypes.byref(pl), ctypes.byref(n)) == -2


@pytest.mark.skipif(not os.path.exists("/root/reference/src/utils/stream_helper.py"), reason="reference tree absent")
def test_writer_matches_the_reference_module(nat):
    spec = importlib.util.spec_from_file_location("ref_stream_helper", "/root/reference/src/utils/stream_helper.py")
    try:
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    except Exception as e:                       # the reference needs Python >= 3.12 syntax in places
        pytest.skip("reference stream_helper does not import here: %s" % e)
    units = _sequence()
    assert _native_bytes(nat, units) == _python_bytes(units, ref)
