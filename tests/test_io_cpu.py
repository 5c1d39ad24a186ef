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


def test_weight_file_round_trip(tmp_path):
    """.dcvw (the native tool's weight file, dcvc_amd/export_weights.py): every tensor set_param() receives - the fp16
    state_dict and the four int32 entropy tables - comes back bit for bit, records are 8-byte aligned."""
    from codec_util import dmc_ld_model
    from dcvc_amd import export_weights
    m = dmc_ld_model(skip_thres=0.15)
    path = str(tmp_path / "ld.dcvw")
    n = export_weights.write_dcvw(path, "ld", m, 0.15)
    kind, thres, tensors = export_weights.read_dcvw(path)
    assert kind == "ld" and abs(thres - 0.15) < 1e-7 and len(tensors) == n
    sd = m.add_cdf_to_state_dict(m.state_dict())
    assert set(tensors) == set(sd)
    for name, t in sd.items():
        a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        want = a.astype(np.float16) if a.dtype.kind == "f" else a
        assert tensors[name].dtype == want.dtype and np.array_equal(tensors[name], want), name
    assert tensors["gaussian_encoder.quantized_cdf"].dtype == np.int32
