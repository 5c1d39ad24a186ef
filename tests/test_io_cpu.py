"""CPU tests of the host-side pieces either side of the hot path: the
stream container and the oracle's picture I/O, against vectors produced by the reference's own
helpers (tests/golden/make_io_golden.py)."""
import io
import os

import numpy as np
import pytest

from dcvc_amd import stream_helper as sh


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "io_golden.npz"))


def test_container_reads_and_rewrites_reference_bytes(golden):
    data = golden["container"].tobytes()
    f = io.BytesIO(data)
    out = io.BytesIO()
    for sid, h, w in golden["sps_cases"]:
        hd = sh.read_header(f)
        assert hd["nal_type"] == sh.NalType.NAL_SPS and hd["sps_id"] == sid
        sps = sh.read_sps_remaining(f, hd["sps_id"])
        assert (sps["height"], sps["width"]) == (h, w)
        sh.write_sps(out, sps)
    for is_i, sid, qp, ec, reset, n in golden["ip_cases"]:
        hd = sh.read_header(f)
        assert hd["nal_type"] == (sh.NalType.NAL_I if is_i else sh.NalType.NAL_P) and hd["sps_id"] == sid
        got_qp, got_ec, got_reset, payload = sh.read_ip_remaining(f)
        assert (got_qp, got_ec, got_reset, len(payload)) == (qp, ec, reset, n)
        written = sh.write_ip(out, bool(is_i), int(sid), int(qp), int(ec), int(reset), payload)
        assert written == 3 + (1 if n < 128 else 2 if n < 16384 else 4) + n
    assert f.read() == b""
    assert out.getvalue() == data


def test_container_rejects_bad_input():
    with pytest.raises(ValueError):
        sh.write_uint_adaptive(io.BytesIO(), 1 << 30)
    with pytest.raises(EOFError):
        sh.read_ip_remaining(io.BytesIO(b"\x20\x02\x85"))
    h = sh.SPSHelper()
    assert h.get_sps_id({"height": 1080, "width": 1920}) == (0, True)
    assert h.get_sps_id({"height": 1080, "width": 1920}) == (0, False)
    assert h.get_sps_id({"height": 720, "width": 1280}) == (1, True)
    assert h.get_sps_by_id(1)["width"] == 1280 and h.get_sps_by_id(9) is None


def test_picture_io_oracle_equals_reference_ops(golden):
    from oracle import frame_io
    x = frame_io.yuv420_to_x(golden["y"], golden["uv"])
    assert x.dtype == np.float16 and np.array_equal(x, golden["x"])
    h, w = golden["y"].shape
    r = frame_io.x_to_yuv420(golden["x_hat"], h, w)
    for k in ("y16", "uv16", "y8", "uv8"):
        assert np.array_equal(r[k], golden[k]), k
