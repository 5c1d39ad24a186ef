"""Generates tests/golden/rans_golden.npz from the REFERENCE itself (run in the build container,
where /root/reference exists):

  * byte streams of the compiled reference coder (oracle/_ref, built from
    /root/reference/src/cpp/py_rans/*.cpp) for seeded symbol arrays, every parallelism 1..8,
    sizes incl. empty / ragged, escape-path symbols (|s| > max_value) included;
  * the Gaussian CDF table produced by the reference Python (`GaussianEncoder.update`,
    /root/reference/src/models/entropy_models.py:184-217) and a seeded BitEstimator table
    (entropy_models.py:113-149), both through the reference pmf_to_quantized_cdf.

Usage: python tests/golden/make_rans_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_oracle, rans as orc  # noqa: E402

SIZES = (0, 1, 7, 100, 4099, 70001)


def case_inputs(seed, count):
    """Seeded y symbols / indexes and z symbols; shared with tests/test_rans.py."""
    rng = np.random.default_rng(seed)
    sym = rng.integers(-6, 7, count).astype(np.int16)
    big = rng.random(count) < 0.03
    sym[big] = rng.integers(-128, 128, int(big.sum()))
    idx = rng.integers(0, 128, count).astype(np.int16)
    comb = ((sym << 8) + idx).astype(np.int16)
    z = rng.integers(-64, 64, 128 * 5 + 3).astype(np.int8)
    return comb, z


def main():
    build_oracle.build_ref()
    ref = orc.load_ref()
    assert ref is not None, "needs /root/reference"

    # reference Python on top of the reference coder
    sys.path.insert(0, "/root/reference")
    sys.modules["MLCodec_extensions_cpp"] = ref
    from src.models.entropy_models import BitEstimator, EntropyCoder, GaussianEncoder
    ec = EntropyCoder()
    g = GaussianEncoder()
    g.update(ec, skip_thres=0.0)
    y_cdf, y_len = g.get_cdf_info()
    torch.manual_seed(1234)
    b = BitEstimator(2, 128)
    with torch.no_grad():
        for p in (b.h, b.b, b.a):
            p.mul_(60.0)       # spread the 0.01-sigma init so that table lengths vary
    b.update(ec)
    z_cdf, z_len = b.get_cdf_info()

    out = {
        "y_cdf": np.asarray(y_cdf, np.int32), "y_len": np.asarray(y_len, np.int32),
        "z_cdf": np.asarray(z_cdf, np.int32), "z_len": np.asarray(z_len, np.int32),
        "z_h": b.h.detach().numpy(), "z_b": b.b.detach().numpy(), "z_a": b.a.detach().numpy(),
    }
    digests = []
    for n in range(1, 9):
        for count in SIZES:
            comb, z = case_inputs(1000 * n + count, count)
            e = ref.RansEncoder()
            e.set_cdf(out["z_cdf"], out["z_len"], 0)
            e.set_cdf(out["y_cdf"], out["y_len"], 1)
            e.reset()
            e.set_entropy_coder_parallel(n)
            e.encode_y(comb)
            e.encode_y(comb[::-1].copy())
            e.encode_z(z, 128, 128)
            e.flush()
            s = np.array(e.get_encoded_stream())
            digests.append(hashlib.sha256(s.tobytes()).hexdigest())
            if count <= 100:
                out["stream_n%d_c%d" % (n, count)] = s
    out["digests"] = np.array(digests)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rans_golden.npz"), **out)
    print("wrote rans_golden.npz:", len(digests), "cases; y table", out["y_cdf"].shape,
          "lens", out["y_len"].min(), out["y_len"].max(), "z lens", out["z_len"].min(),
          out["z_len"].max())


if __name__ == "__main__":
    main()
