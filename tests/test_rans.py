"""rANS coder parity: product (libdcvc_amd.so, C ABI) vs the oracle restatement
(oracle/rans_oracle.c) vs golden vectors generated from the reference itself
(tests/golden/make_rans_golden.py) vs - when present - the compiled reference (oracle/_ref).
Integer work: every comparison is bit-exact."""
import hashlib
import os
import sys

import numpy as np
import pytest

import dcvc_amd
from oracle import rans as orc

dcvc_amd.install_plugin()
import MLCodec_extensions_cpp as mine  # noqa: E402  (the product's plugin module)

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_rans_golden import SIZES, case_inputs  # noqa: E402


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "rans_golden.npz"))


def _product_encode(g, n, comb, z):
    e = mine.RansEncoder()
    e.set_cdf(g["z_cdf"], g["z_len"], 0)
    e.set_cdf(g["y_cdf"], g["y_len"], 1)
    e.reset()
    e.set_entropy_coder_parallel(n)
    e.encode_y(comb)
    e.encode_y(comb[::-1].copy())
    e.encode_z(z, 128, 128)
    e.flush()
    return e.get_encoded_stream()


def _oracle_tables(g):
    t = orc.Tables()
    t.set_cdf(g["z_cdf"], g["z_len"], 0)
    t.set_cdf(g["y_cdf"], g["y_len"], 1)
    return t


def test_streams_match_reference_golden(golden):
    """Product and oracle reproduce the reference's byte streams for all 8 parallelism levels,
    empty / ragged sizes and escape-coded symbols."""
    t = _oracle_tables(golden)
    k = 0
    for n in range(1, 9):
        for count in SIZES:
            comb, z = case_inputs(1000 * n + count, count)
            want = golden["digests"][k]
            k += 1
            s_prod = _product_encode(golden, n, comb, z)
            s_orc = orc.encode(t, [("y", comb), ("y", comb[::-1].copy()), ("z", z, 128, 128)], n)
            assert hashlib.sha256(s_prod.tobytes()).hexdigest() == want, (n, count)
            assert hashlib.sha256(s_orc.tobytes()).hexdigest() == want, (n, count)
            key = "stream_n%d_c%d" % (n, count)
            if key in golden:
                assert np.array_equal(s_prod, golden[key])


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
def test_decode_round_trip(golden, n):
    """encode -> decode closure, decode order z, y(last pushed), y(first pushed), for product and
    oracle decoders on the same reference-identical stream."""
    comb, z = case_inputs(77 + n, 20011)
    comb2 = comb[::-1].copy()
    s = _product_encode(golden, n, comb, z)
    d = mine.RansDecoder()
    d.set_cdf(golden["z_cdf"], golden["z_len"], 0)
    d.set_cdf(golden["y_cdf"], golden["y_len"], 1)
    d.set_entropy_coder_parallel(n)
    d.set_stream(s)
    od = orc.Decoder(_oracle_tables(golden), s, n)
    d.decode_z(z.size, 128, 128)
    assert np.array_equal(d.get_decoded_tensor(), z)
    assert np.array_equal(od.decode_z(z.size, 128, 128), z)
    for c in (comb2, comb):
        idx = (c & 0xff).astype(np.uint8)
        want = (c >> 8).astype(np.int8)
        d.decode_y(idx)
        assert np.array_equal(d.get_decoded_tensor(), want)
        assert np.array_equal(od.decode_y(idx), want)


def test_extreme_symbols(golden):
    """Every int8 value through the narrowest and the widest table (escape path, bypass groups,
    unary group-count continuation)."""
    lens = golden["y_len"]
    for idx in (int(np.argmin(lens)), int(np.argmax(lens))):
        sym = np.arange(-128, 128, dtype=np.int16)
        comb = ((sym << 8) + idx).astype(np.int16)
        z = np.array([-64, 63, 0, 1, -1], dtype=np.int8)
        for n in (1, 2, 3):
            s = _product_encode(golden, n, comb, z)
            s2 = orc.encode(_oracle_tables(golden),
                            [("y", comb), ("y", comb[::-1].copy()), ("z", z, 128, 128)], n)
            assert np.array_equal(s, s2)
            od = orc.Decoder(_oracle_tables(golden), s, n)
            assert np.array_equal(od.decode_z(z.size, 128, 128), z)
            od.decode_y(np.full(256, idx, np.uint8))
            assert np.array_equal(od.decode_y(np.full(256, idx, np.uint8)), sym.astype(np.int8))


def test_pmf_to_quantized_cdf_matches_golden_tables(golden):
    """The product's pmf_to_quantized_cdf rebuilds the reference's Gaussian table
    (entropy_models.py:184-217 restated with numpy/scipy-free math) row by row."""
    import math
    import torch
    scale_table = torch.exp(torch.linspace(math.log(0.11), math.log(16.0), 128))
    # restated from GaussianEncoder.update: symmetric range where cdf(i) <= 0.999, max 8
    for row in (0, 17, 64, 127):
        s = scale_table[row]
        nd = torch.distributions.normal.Normal(0.0, s)
        rng_ = 8
        for i in range(8, 1, -1):
            if nd.cdf(torch.tensor(float(i))) > 0.999:
                rng_ = i
        samples = torch.arange(2 * rng_ + 1).float() - rng_
        pmf = nd.cdf(samples + 0.5) - nd.cdf(samples - 0.5)
        tail = 2 * nd.cdf(samples[:1] - 0.5)
        prob = torch.cat((pmf, tail))
        # reorder to 0, +1, -1, +2, ... (entropy_models.py:45-57)
        center = (prob.numel() - 1) // 2
        re = prob.clone()
        re[0] = prob[center]
        for i in range(1, center + 1):
            re[2 * i - 1] = prob[center + i]
            re[2 * i] = prob[center - i]
        cdf = mine.pmf_to_quantized_cdf(re.tolist())
        assert cdf == orc.pmf_to_quantized_cdf(re.numpy())
        assert golden["y_len"][row] == len(cdf)
        assert list(golden["y_cdf"][row][:len(cdf)]) == cdf


def test_against_compiled_reference_if_present(golden):
    """Live cross-check with the reference coder built from its own sources (oracle/_ref)."""
    ref = orc.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 8):
        comb, z = case_inputs(4242 + n, int(rng.integers(1, 50000)))
        e = ref.RansEncoder()
        e.set_cdf(golden["z_cdf"], golden["z_len"], 0)
        e.set_cdf(golden["y_cdf"], golden["y_len"], 1)
        e.reset()
        e.set_entropy_coder_parallel(n)
        e.encode_y(comb)
        e.encode_y(comb[::-1].copy())
        e.encode_z(z, 128, 128)
        e.flush()
        assert np.array_equal(np.array(e.get_encoded_stream()), _product_encode(golden, n, comb, z))


@pytest.mark.parametrize("avx512", ["0", "1"])
def test_both_cdf_search_paths(avx512):
    """The decoder's CDF search has a portable form (start table + scan) and an AVX-512 form (one compare + popcount);
    which one runs is decided once per process (host capability, DCVC_RANS_AVX512=0 forces the portable one). Every
    decode test of this file again in a child process with the switch at 0 and at 1: on an AVX-512 host both forms see
    the golden vectors, elsewhere the portable form runs twice."""
    import subprocess
    import sys
    if os.environ.get("DCVC_RANS_CHILD"):
        pytest.skip("already in the child process")
    env = dict(os.environ, DCVC_RANS_AVX512=avx512, DCVC_RANS_CHILD="1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k",
                          "decode_round_trip or extreme_symbols or reference_golden or zero_frequency"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and " passed" in res.stdout, res.stdout[-3000:] + res.stderr[-1000:]


def test_zero_frequency_leading_symbol_decodes_the_same_on_both_paths():
    """A CDF row with cdf[1] == 0 (symbol 0 has frequency 0) is outside what the vector search can represent (cdf[j] - 1
    wraps): CdfTable::load must keep the portable search for such a table, so a stream coded with it decodes to the
    symbols that went in whatever the host."""
    y_cdf = np.zeros((2, 8), np.int32)
    y_cdf[0, :6] = [0, 0, 20000, 50000, 65535, 65536]          # symbol 0 never occurs, 4 coded values + the escape slot
    y_cdf[1, :5] = [0, 30000, 60000, 65535, 65536]
    y_len = np.array([6, 5], np.int32)
    z_cdf = np.zeros((128, 4), np.int32)
    z_cdf[:, :4] = [0, 30000, 65535, 65536]
    z_len = np.full(128, 4, np.int32)
    rng = np.random.default_rng(3)
    # values whose interleaved code (0, +1, -1, +2, ...) is >= 1 for table 0: never the zero-frequency symbol
    sym = rng.choice(np.array([1, -1, 2], np.int16), size=4000)
    idx = np.zeros(4000, np.int16)
    comb = ((sym << 8) + idx).astype(np.int16)
    e = mine.RansEncoder()
    e.set_cdf(z_cdf, z_len, 0)
    e.set_cdf(y_cdf, y_len, 1)
    e.reset()
    e.set_entropy_coder_parallel(1)
    e.encode_y(comb)
    e.flush()
    s = np.array(e.get_encoded_stream())
    d = mine.RansDecoder()
    d.set_cdf(z_cdf, z_len, 0)
    d.set_cdf(y_cdf, y_len, 1)
    d.set_entropy_coder_parallel(1)
    d.set_stream(s)
    d.decode_y(idx.astype(np.uint8))
    assert np.array_equal(d.get_decoded_tensor(), sym.astype(np.int8))


def test_errors_are_reported():
    e = mine.RansEncoder()
    with pytest.raises(RuntimeError):
        e.set_entropy_coder_parallel(9)
    with pytest.raises(RuntimeError):
        e.set_cdf(np.zeros((2, 4), np.int32), np.array([3, 3], np.int32), 2)
