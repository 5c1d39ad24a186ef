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
        # north_star's tolerance: the picture quality (PSNR against the source) of the fp16 matrix-core path is within
        # 0.02 dB of the reference's fp32 graph
        src = x.astype(np.float32)
        assert abs(psnr(r["x_hat"], src) - psnr(ref, src)) <= 0.02


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


# ------------------------------------------------------------------------------ inter model (LD)
@pytest.fixture(scope="module")
def golden_ld(golden_dir):
    return np.load(os.path.join(golden_dir, "dmcld_golden.npz"))


def test_ld_parameter_inventory():
    from dcvc_amd import arch
    spec = arch.dmc_ld_spec()
    assert arch.param_count(spec) == 9650176           # strict=True load in make_dmcld_golden.py
    assert spec["temporal_prior_encoder.conv.down.weight"] == (256, 1024, 1, 1)
    assert spec["decoder.up.conv.0.weight"] == (1024, 128, 1, 1)


def test_ld_oracle_follows_reference_graph(golden_ld):
    """Every P picture on its own, started from the reference graph's temporal state (memory +
    reference feature after the previous picture), incl. the picture after a memory reset: the
    reconstruction must equal the reference's fp32 forward_one_frame up to fp16 noise."""
    from codec_util import dmc_ld_model, force_ld_state
    from oracle import codec
    g = golden_ld
    m = dmc_ld_model()
    for s in range(2):
        plan = g["s%d_plan" % s]
        for i in range(len(plan)):
            qp, reset = int(plan[i][0]), bool(plan[i][1])
            o = codec.DMCLDOracle(m.state_dict(), -60000.0, m.get_cdf_info())
            if i == 0:
                o.add_ref_feature_from_frame(g["s%d_ref" % s], True)
            else:
                key = "s%d_mem%d" % (s, i - 1)
                force_ld_state(o, g["s%d_feat%d" % (s, i - 1)], g[key] if key in g else None)
            o.compress(g["s%d_x%d" % (s, i)], qp, reset)
            _, x_hat = o.recon_head(o.feature_p)
            ref = np.clip(g["s%d_xhat%d" % (s, i)].astype(np.float32), -0.5, 0.5)
            p = psnr(x_hat, ref)
            src = g["s%d_x%d" % (s, i)].astype(np.float32)
            h, w = src.shape[:2]
            dp = abs(psnr(x_hat[:h, :w], src) - psnr(ref[:h, :w], src))
            print("seq", s, "picture", i, "PSNR oracle vs reference graph: %.2f dB, |delta PSNR vs source| %.4f dB" % (p, dp))
            assert p > 47.0
            # north_star's tolerance on the reconstructions: within 0.02 dB PSNR of the reference's fp32 graph
            assert dp <= 0.02


def test_ld_oracle_sequence_closure(golden_ld):
    """Free-running encoder and decoder oracles over a sequence with a reset: identical temporal
    state on both sides, decoded pictures track the reference graph."""
    from codec_util import dmc_ld_model, oracle_for
    g = golden_ld
    m = dmc_ld_model(skip_thres=0.15)
    enc, dec = oracle_for(m), oracle_for(m)
    ref = g["s0_ref"]
    enc.add_ref_feature_from_frame(ref, True)
    dec.add_ref_feature_from_frame(ref, False)
    for i, (qp, reset) in enumerate(g["s0_plan"]):
        x = g["s0_x%d" % i]
        r = enc.compress(x, int(qp), bool(reset))
        xd = dec.decompress(r["bit_stream"], int(qp), x.shape[0], x.shape[1], r["ec_parallel"], bool(reset))
        assert np.array_equal(dec.feature_p, enc.feature_p)
        _, xe = enc.recon_head(enc.feature_p)
        assert np.array_equal(xd, xe)
        # skip mode (inference only) zeroes low-scale symbols, so only a loose bound holds here
        p = psnr(xd, np.clip(g["s0_xhat%d" % i].astype(np.float32), -0.5, 0.5))
        print("picture", i, "bytes", len(r["bit_stream"]), "PSNR vs graph %.2f dB" % p)
        assert p > 20.0


# ------------------------------------------------------------------------------ inter models (HT-S / HT-L)
@pytest.fixture(scope="module")
def golden_ht(golden_dir):
    return np.load(os.path.join(golden_dir, "dmcht_golden.npz"))


def test_ht_parameter_inventory():
    from dcvc_amd import arch
    assert arch.param_count(arch.dmc_ht_spec(True)) == 81157632       # strict=True load in make_dmcht_golden.py
    assert arch.param_count(arch.dmc_ht_spec(False)) == 120538368
    assert arch.dmc_ht_spec(False)["decoder.up.conv.0.weight"] == (2048, 256, 3, 3)
    assert "hyper_decoder.conv.0.up.conv.0.bias" not in arch.dmc_ht_spec(True)


@pytest.mark.parametrize("structure,chunks,floor", [("hts", (0, 1), 42.0), ("htl", (1, 2), 52.0)])
def test_ht_oracle_follows_reference_graph(golden_ht, structure, chunks, floor):
    """Single chunks (8 pictures) started from the reference graph's temporal state, incl. a
    chunk that resets the memory and the chunk after it. The HT-S floor is lower: its one-shot
    4-step scheme turns every rounding-tie flip of an early step into shifted means later on
    (measured: ~1 % flipped symbols, everything else within fp16 noise)."""
    from codec_util import dmc_ht_model, force_ld_state
    from oracle import codec
    g = golden_ht
    m = dmc_ht_model(structure)
    plan = g[structure + "_plan"]
    for i in chunks:
        qp, reset = int(plan[i][0]), bool(plan[i][1])
        o = codec.DMCHTOracle(m.state_dict(), -60000.0, m.get_cdf_info(), is_hts=m.is_hts)
        if i == 0:
            o.add_ref_feature_from_frame(g[structure + "_ref"], True)
        else:
            key = "%s_mem%d" % (structure, i - 1)
            force_ld_state(o, g["%s_feat%d" % (structure, i - 1)], g[key] if key in g else None)
        o.compress(g["%s_x%d" % (structure, i)], qp, reset)
        x_hat = np.concatenate(o.recon_head(o.feature_p)[0], axis=-1)
        ref = np.clip(g["%s_xhat%d" % (structure, i)].astype(np.float32), -0.5, 0.5)
        p = psnr(x_hat, ref)
        src = g["%s_x%d" % (structure, i)].astype(np.float32)
        h, w = src.shape[:2]
        dp = abs(psnr(x_hat[:h, :w], src) - psnr(ref[:h, :w], src))
        print(structure, "chunk", i, "PSNR oracle vs reference graph: %.2f dB, |delta PSNR vs source| %.4f dB" % (p, dp))
        assert p > floor
        assert dp <= 0.02        # north_star's tolerance (measured <= 0.0045 dB over all fixture chunks)


def test_ht_oracle_sequence_closure(golden_ht):
    """Free-running HT-S encoder and decoder oracles over two chunks with a reset in between."""
    from codec_util import dmc_ht_model, oracle_for
    g = golden_ht
    m = dmc_ht_model("hts", skip_thres=0.15)
    enc, dec = oracle_for(m), oracle_for(m)
    enc.add_ref_feature_from_frame(g["hts_ref"], True)
    dec.add_ref_feature_from_frame(g["hts_ref"], False)
    for i, reset in ((0, True), (1, False)):
        x = g["hts_x%d" % i]
        r = enc.compress(x, 30, reset)
        xd = dec.decompress(r["bit_stream"], 30, x.shape[0], x.shape[1], r["ec_parallel"], reset)
        assert len(xd) == 8 and xd[0].shape == (64, 64, 3)
        assert np.array_equal(dec.feature_p, enc.feature_p)
        assert np.array_equal(np.concatenate(xd, -1), np.concatenate(enc.recon_head(enc.feature_p)[0], -1))


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


def _vec_lib():
    import ctypes
    from oracle import nn
    lib = nn._lib()
    lib.orc_avx512.restype = ctypes.c_int
    lib.orc_avx512.argtypes = [ctypes.c_int]
    lib.orc_mfma16_vec.restype = ctypes.c_float
    lib.orc_mfma16_vec.argtypes = [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def test_vector_contraction_matches_hardware_measurements(golden_dir):
    """The AVX-512 restatement of the contraction arithmetic (what the oracle runs on hosts that
    have it) against the same MI355X probe data, and against the scalar routine."""
    import ctypes
    lib = _vec_lib()
    if not lib.orc_avx512(-1):
        pytest.skip("host without AVX-512: the oracle runs the scalar routine")
    d = np.load(os.path.join(golden_dir, "mfma_probe.npz"))
    a, b, c, out = d["a"], d["b"], d["c"], d["d"]
    bad = 0
    for i in range(len(c)):
        ai, bi = np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i])
        w = np.float32(lib.orc_mfma16_vec(ctypes.c_float(float(c[i])), ai.ctypes.data, bi.ctypes.data))
        if w.tobytes() != out[i].tobytes() and not (w == 0 and out[i] == 0):
            bad += 1
    assert bad == 0, "%d of %d trials differ" % (bad, len(c))


@pytest.mark.parametrize("kind", ["normal", "wide", "sparse", "bits"])
def test_vector_and_scalar_contractions_are_identical(kind):
    from oracle import nn
    lib = _vec_lib()
    if not lib.orc_avx512(-1):
        pytest.skip("host without AVX-512")
    rng = np.random.default_rng(7)

    def draw(shape):
        if kind == "normal":
            return (rng.standard_normal(shape) * 0.2).astype(np.float16)
        if kind == "wide":       # exponents spread over the whole fp16 range incl. subnormals
            return (rng.standard_normal(shape) * np.exp2(rng.integers(-24, 4, size=shape))).astype(np.float16)
        if kind == "sparse":
            x = rng.standard_normal(shape).astype(np.float16)
            x[rng.random(shape) < 0.6] = 0
            return x
        bits = rng.integers(0, 0x7C00, size=shape).astype(np.uint16) | (rng.integers(0, 2, size=shape).astype(np.uint16) << 15)
        return bits.view(np.float16)     # every finite bit pattern (outputs may overflow: compared as bits)

    try:
        for P, K, N, wsilu, chunk in ((37, 64, 64, False, False), (19, 128, 256, True, True), (9, 384, 64, True, False)):
            x, w, b, r = draw((P, K)), draw((N, K)), draw((N,)), draw((P, N))
            res = None if chunk else r
            lib.orc_avx512(0)
            y0 = nn.conv1x1(x, w, b, r1=res, wsilu=wsilu, chunk_add=chunk)
            lib.orc_avx512(1)
            y1 = nn.conv1x1(x, w, b, r1=res, wsilu=wsilu, chunk_add=chunk)
            assert np.array_equal(y0.view(np.uint16), y1.view(np.uint16))
    finally:
        lib.orc_avx512(1)


def test_torch_restatement_of_the_reference_cpu_path(golden):
    """oracle/torch_graph.py (what bench.py times as the reference's CPU path on the GPU box, where
    /root/reference does not exist) against x_hat made by the reference's own
    DMCI.forward_one_frame (tests/golden/make_dmci_golden.py), and live against the reference module
    when the tree is present. fp32 graph vs fp32 graph: equal up to summation order."""
    import torch
    from dcvc_amd import arch, synthetic
    from oracle import torch_graph
    sd = synthetic.synthetic_state_dict(arch.dmci_spec(), 0)
    for i, qp in enumerate((32, 5, 60)):
        x = torch.from_numpy(golden["x_%d" % i].astype(np.float32)).permute(2, 0, 1).unsqueeze(0)
        got = torch_graph.forward_one_frame(sd, x, qp)[0].permute(1, 2, 0).numpy()
        want = golden["xhat_%d" % i].astype(np.float32)
        # a flipped rounding decision moves a latent by one step: bound the damage, demand near-identity
        psnr = 10 * np.log10(1.0 / np.mean((got - want) ** 2))
        assert psnr > 55, (i, psnr)
    if os.path.isdir("/root/reference/src/models"):
        import sys
        from oracle import build_oracle, rans as orc
        try:
            build_oracle.build_ref()
            sys.modules.setdefault("MLCodec_extensions_cpp", orc.load_ref())
            sys.path.insert(0, "/root/reference")
            from src.models.image_model import DMCI
        except Exception as e:
            pytest.skip("reference module does not import here: %s" % e)
        finally:
            if "/root/reference" in sys.path:
                sys.path.remove("/root/reference")
        net = DMCI().eval()
        net.load_state_dict(sd, strict=True)
        x = torch.rand((1, 3, 64, 128), generator=torch.Generator().manual_seed(3)) - 0.5
        with torch.no_grad():
            want = net.forward_one_frame(x, torch.tensor([20]), recon_only=True)
        got = torch_graph.forward_one_frame(sd, x, 20)
        assert torch.allclose(got, want, atol=2e-5), float((got - want).abs().max())


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


def test_symbol_math_matches_the_references_python_fallbacks(golden_dir):
    """oracle/symbols_np.py against vectors made with DCVC-RT's PyTorch fallbacks of the same kernels
    (DCVC-family/DCVC-RT/src/layers/cuda_inference.py:26-33,58-74,113-119,124-171; generator:
    tests/golden/make_symbols_golden.py), on inputs without rounding ties. Bit-exact."""
    from oracle import symbols_np as sym
    g = np.load(os.path.join(golden_dir, "symbols_rt_golden.npz"))
    for case in range(4):
        y, means, scales, thres = g["y%d" % case], g["means%d" % case], g["scales%d" % case], float(g["thres%d" % case])
        H, W, C = y.shape
        for k, mask in enumerate(sym.get_mask_4x(H, W, C)):
            tag = "%d_%d" % (case, k)
            y_q, y_hat, s_hat = sym.process_with_mask(y, scales, means, mask, thres)
            for name, got in (("y_q", y_q), ("y_hat", y_hat), ("s_hat", s_hat)):
                assert np.array_equal(got, g[name + tag]), (name, tag)    # values (-0.0 == 0.0: the sign of zero is not coded)
            s_w, y_w = sym.fold4(s_hat), sym.fold4(y_q)
            idx, keep = sym.build_index_dec(s_w, thres)
            assert np.array_equal(keep.reshape(H, W, C // 4), g["keep" + tag])
            assert np.array_equal(idx.reshape(H, W, C // 4), g["idx" + tag])
            comb, keep_e = sym.build_index_enc(y_w, s_w, thres)
            assert np.array_equal(keep_e, keep)
            assert np.array_equal(comb[keep_e], g["comb" + tag])
            if k == 0:
                assert np.array_equal(sym.restore_y_4x(sym.fold4(y_q), means, mask), g["restored" + tag])
        z_hat, z_i8 = sym.round_z(g["z%d" % case])
        assert np.array_equal(z_hat, g["z_hat%d" % case]) and np.array_equal(z_i8, g["z_i8%d" % case])
