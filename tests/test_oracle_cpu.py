"""CPU tests (no GPU): the oracle against the golden vectors generated from the reference
(tests/golden/make_*_golden.py), host logic, and the C-ABI surface."""
import os

import numpy as np
import pytest

from codec_util import dmci_model, oracle_for, psnr


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "dmci_golden.npz"))


def test_parameter_inventory_matches_reference():
    from dcvc_amd import arch
    spec = arch.dmci_spec()
    assert arch.param_count(spec) == 42179328          # 42.2 M, SURVEY §2.1 / assets/complexity.png
    assert spec["enc.enc_2.6.weight"] == (256, 384, 3, 3)
    assert spec["dec.dec_1.0.up.conv.0.weight"] == (1536, 256, 1, 1)
    assert "dec.dec_1.0.up.conv.0.bias" not in spec


def test_cdf_tables_equal_reference(golden):
    """dcvc_amd.models restates GaussianEncoder.update / BitEstimator.update; the tables must be
    the reference's, entry for entry."""
    m = dmci_model()
    z_cdf, z_len, y_cdf, y_len = m.get_cdf_info()
    assert np.array_equal(y_cdf, golden["y_cdf"]) and np.array_equal(y_len, golden["y_len"])
    assert np.array_equal(z_len, golden["z_len"])
    assert np.array_equal(z_cdf[:512], golden["z_cdf_head"])
    assert int(z_cdf.astype(np.int64).sum()) == int(golden["z_cdf_sum"][0])


def test_oracle_follows_reference_graph(golden):
    """With the skip mode disabled (the training graph has none) the oracle's
    reconstruction equals the reference's fp32 forward_one_frame up to fp16 storage noise and the
    occasional rounding-tie symbol flip of an (untrained, chaotic) random network."""
    from oracle import codec
    m = dmci_model()
    o = codec.DMCIOracle(m.state_dict(), -60000.0, m.get_cdf_info())
    for i, qp in enumerate((32, 5)):
        x = golden["x_%d" % i]
        r = o.compress(x, qp)
        ref = np.clip(golden["xhat_%d" % i].astype(np.float32), -0.5, 0.5)
        p = psnr(r["x_hat"], ref)
        print("case", i, "PSNR oracle vs reference graph: %.2f dB" % p)
        assert p > 45.0


@pytest.mark.parametrize("hw,qp", [((64, 64), 32), ((40, 72), 0)])
def test_oracle_closure(hw, qp):
    """encode -> decode closure of the oracle itself, padded and unpadded sizes."""
    from codec_util import picture
    m = dmci_model(skip_thres=0.15)
    o = oracle_for(m)
    x = picture(*hw)
    r = o.compress(x, qp)
    assert len(r["bit_stream"]) > 0
    xd = o.decompress(r["bit_stream"], qp, hw[0], hw[1], r["ec_parallel"])
    assert np.array_equal(xd, r["x_hat"])
    assert r["x_hat"].shape == ((hw[0] + 15) // 16 * 16, (hw[1] + 15) // 16 * 16, 3)


def test_mfma_model_matches_hardware_measurements(golden_dir):
    """The oracle's contraction arithmetic against v_mfma_f32_32x32x16_f16 outputs recorded on an
    MI355X (tools/mfma_probe2.hip): every trial bit-exact."""
    from oracle import nn
    d = np.load(os.path.join(golden_dir, "mfma_probe.npz"))
    a, b, c, out = d["a"], d["b"], d["c"], d["d"]
    bad = 0
    for i in range(len(c)):
        w = nn.mfma16(c[i], a[i], b[i])
        if w.tobytes() != out[i].tobytes() and not (w == 0 and out[i] == 0):
            bad += 1
    assert bad == 0, "%d of %d trials differ" % (bad, len(c))


def test_c_abi_exports_every_declared_symbol():
    from tools import check_abi
    assert len(check_abi.declared_symbols()) >= 40
    assert check_abi.missing_symbols() == []


def test_compress_without_gpu_fails_loudly():
    """No CPU fallback: on a machine without a GPU the product path raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dcvc_amd
    dcvc_amd.install_plugin()
    import inference_extensions_cuda as ext
    with pytest.raises(Exception):
        p = ext.DMCIProxy()
        p.compress(torch.zeros((1, 3, 64, 64), dtype=torch.float16), 0, 0, 0)
