"""Host entropy-coder throughput on symbols shaped like a 1080p intra picture (f3: the decode path's host share).
y symbols of DCVC-UF: int16 (symbol << 8) + scale index, scale index ~ log-uniform, symbols ~ rounded Gaussian of
that scale; 4 coding steps per picture. Prints M symbols/s for encode and decode at ec_parallel 1 and 8."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import dcvc_amd  # noqa: E402
from dcvc_amd import arch, models, synthetic  # noqa: E402

dcvc_amd.install_plugin()
import MLCodec_extensions_cpp as ec  # noqa: E402


def main():
    n_per_step = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
    spread = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0        # > 1: escape-coded symbols (the low-qp streams)
    net = models.DMCI()
    net.load_state_dict(synthetic.synthetic_state_dict(arch.dmci_spec(), 0))
    net.update(0.15)
    z_cdf, z_len, y_cdf, y_len = net.get_cdf_info()
    rng = np.random.default_rng(1)
    steps = []
    for _ in range(4):
        idx = rng.integers(8, 100, n_per_step)
        scale = np.exp(-2.2073 + idx * (2.7726 + 2.2073) / 127)
        sym = np.clip(np.rint(rng.standard_normal(n_per_step) * scale * spread), -127, 127).astype(np.int64)
        steps.append((((sym << 8) + idx).astype(np.int16), idx.astype(np.uint8)))
    for par in (1, 8):
        e = ec.RansEncoder()
        e.set_cdf(z_cdf, z_len, 0)
        e.set_cdf(y_cdf, y_len, 1)
        e.set_entropy_coder_parallel(par)
        best_e = best_d = 1e9
        for _ in range(5):
            e.reset()
            t0 = time.perf_counter()
            for comb, _ in steps:
                e.encode_y(comb)
            e.flush()
            stream = e.get_encoded_stream()
            best_e = min(best_e, time.perf_counter() - t0)
        d = ec.RansDecoder()
        d.set_cdf(z_cdf, z_len, 0)
        d.set_cdf(y_cdf, y_len, 1)
        d.set_entropy_coder_parallel(par)
        for _ in range(5):
            d.set_stream(stream)
            t0 = time.perf_counter()
            outs = []
            for _, idx in reversed(steps):            # calls decode last-in first-out
                d.decode_y(idx)
                outs.append(d.get_decoded_tensor().copy())
            best_d = min(best_d, time.perf_counter() - t0)
        for (comb, _), out in zip(reversed(steps), outs):
            assert np.array_equal(out.astype(np.int16), comb >> 8)
        n = 4 * n_per_step
        print("ec_parallel %d: %d symbols, %d bytes; encode %.2f ms (%.0f Msym/s), decode %.2f ms (%.0f Msym/s)"
              % (par, n, len(stream), 1e3 * best_e, n / best_e / 1e6, 1e3 * best_d, n / best_d / 1e6))


if __name__ == "__main__":
    main()
