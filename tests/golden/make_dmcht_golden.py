"""Generates tests/golden/dmcht_golden.npz from the REFERENCE Python model (run in the build
container, where /root/reference exists):

  * loads the seeded synthetic weights into the reference hierarchical ``DMC`` for both
    structures (/root/reference/src/models/video_model_ht.py:320, ModelStructure.HTS / HTL) with
    strict=True - proves that dcvc_amd/arch.py:dmc_ht_spec is the reference's parameter inventory;
  * runs the reference's only CPU-runnable path, the fp32 graph ``forward_one_frame``
    (video_model_ht.py:446-492; one call = a chunk of 8 pictures), over short sequences
    including a feature-memory reset, seeded through the training-mode
    ``add_ref_feature_from_frame`` -> the 8 reconstructions per chunk and the temporal state
    (memory, reference feature) after each chunk.

Usage: python tests/golden/make_dmcht_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dcvc_amd import arch, synthetic  # noqa: E402
from oracle import build_oracle, rans as orc  # noqa: E402

H, W = 64, 64          # the graph needs multiples of 64
PLANS = {"hts": [(32, 0), (40, 1), (12, 0)], "htl": [(32, 0), (50, 1), (20, 0)]}   # (qp, reset) per chunk
SEED = 0


def picture(index, seed):
    yy, uv = synthetic.synthetic_frame_yuv420(H, W, index, seed)
    return synthetic.yuv420_to_x(yy, uv)


def hwc(t):
    return t[0].permute(1, 2, 0).numpy().astype(np.float16)


def main():
    build_oracle.build_ref()
    ref = orc.load_ref()
    sys.path.insert(0, "/root/reference")
    sys.modules["MLCodec_extensions_cpp"] = ref
    from src.models.video_model_ht import DMC
    from src.utils.common import ModelStructure
    torch.set_num_threads(8)
    out = {}
    for name, plan in PLANS.items():
        hts = name == "hts"
        net = DMC(ModelStructure.HTS if hts else ModelStructure.HTL)
        net.load_state_dict(synthetic.synthetic_state_dict(arch.dmc_ht_spec(hts), SEED), strict=True)
        net.train()           # selects the graph-only branch of add_ref_feature_from_frame
        with torch.no_grad():
            ref_frame = picture(0, SEED).half().float()
            out[name + "_ref"] = hwc(ref_frame)
            net.add_ref_feature_from_frame(ref_frame)
            for i, (qp, reset) in enumerate(plan):
                x = torch.cat([picture(1 + 8 * i + j, SEED) for j in range(8)], dim=1).half().float()
                r = net.forward_one_frame(x, torch.tensor([qp]), reset_feature_memory=bool(reset))
                x_hat = torch.cat(r["x_hat"], dim=1)
                out["%s_x%d" % (name, i)] = hwc(x)
                out["%s_xhat%d" % (name, i)] = hwc(x_hat)
                out["%s_feat%d" % (name, i)] = hwc(net.ref_feature)
                if net.memory is not None:
                    out["%s_mem%d" % (name, i)] = hwc(net.memory)
                print(name, "chunk", i, (qp, reset), "x_hat std %.4f" % x_hat.std().item())
        out[name + "_plan"] = np.array(plan, dtype=np.int32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dmcht_golden.npz"), **out)
    print("wrote dmcht_golden.npz")


if __name__ == "__main__":
    main()
