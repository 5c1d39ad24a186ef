"""Generates tests/golden/dmci_golden.npz from the REFERENCE Python model (run in the build
container, where /root/reference exists):

  * loads the seeded synthetic weights (dcvc_amd/synthetic.py) into the reference ``DMCI``
    (/root/reference/src/models/image_model.py:126) with strict=True - proves that the parameter
    inventory in dcvc_amd/arch.py is the reference's;
  * runs the reference ``forward_one_frame(x, qp, recon_only=True)`` (image_model.py:150-170, the
    only CPU-runnable path of the reference) on small synthetic pictures -> x_hat (fp32 graph);
  * runs the reference ``update()`` (common_model.py:152-155) -> CDF tables of both entropy
    models, against which dcvc_amd/models.py's restatement is checked.

Usage: python tests/golden/make_dmci_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dcvc_amd import arch, synthetic  # noqa: E402
from oracle import build_oracle, rans as orc  # noqa: E402

CASES = [(64, 64, 32), (64, 128, 5), (128, 64, 60)]   # (H, W, qp); the training graph needs multiples of 64
SEED = 0


def main():
    build_oracle.build_ref()
    ref = orc.load_ref()
    sys.path.insert(0, "/root/reference")
    sys.modules["MLCodec_extensions_cpp"] = ref
    from src.models.image_model import DMCI
    torch.set_num_threads(8)
    net = DMCI().eval()
    net.load_state_dict(synthetic.synthetic_state_dict(arch.dmci_spec(), SEED), strict=True)
    net.update(0.0)
    y_cdf, y_len = net.gaussian_encoder.get_cdf_info()
    z_cdf, z_len = net.bit_estimator_z.get_cdf_info()
    out = {"y_cdf": y_cdf.astype(np.int32), "y_len": y_len.astype(np.int32),
           "z_cdf_head": z_cdf[:512].astype(np.int32), "z_len": z_len.astype(np.int32),
           "z_cdf_sum": np.array([int(z_cdf.astype(np.int64).sum())])}
    with torch.no_grad():
        for i, (H, W, qp) in enumerate(CASES):
            yy, uv = synthetic.synthetic_frame_yuv420(H + H % 2, W + W % 2, i, SEED)
            x = synthetic.yuv420_to_x(yy, uv)[:, :, :H, :W]
            pr, pb = DMCI.get_padding_size(H, W, 16)
            xp = torch.nn.functional.pad(x, (0, pr, 0, pb), mode="replicate")
            x_hat = net.forward_one_frame(xp, torch.tensor([qp]), recon_only=True)
            out["x_%d" % i] = x[0].permute(1, 2, 0).numpy().astype(np.float16)
            out["xhat_%d" % i] = x_hat[0].permute(1, 2, 0).numpy().astype(np.float16)
            print("case", i, (H, W, qp), "x_hat std %.4f" % x_hat.std().item())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dmci_golden.npz"), **out)
    print("wrote dmci_golden.npz")


if __name__ == "__main__":
    main()
