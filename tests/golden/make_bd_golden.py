#!/usr/bin/env python
"""Fixture for tests/test_bd_rate.py: the UVG and HEVC_B classes of the reference's VTM-17.0 anchor
(/root/reference/anchors/vtm_17.0_yuv420_LB_allf_ip0.json, keys the BD-rate needs only) plus
known answers computed HERE by an independent restatement of the published metric on top of
scipy.interpolate.pchip_interpolate (the routine the reference's `bd_metric` dependency calls):
  * per-sequence and per-class BD-rate of a pseudo codec = the anchor with its even rate points
    (qp 22, 28, ...) against the anchor's odd ones, and of the anchor with 10 % fewer bits."""
import json
import os

import numpy as np
import scipy.interpolate

SRC = "/root/reference/anchors/vtm_17.0_yuv420_LB_allf_ip0.json"
HERE = os.path.dirname(os.path.abspath(__file__))
KEEP = ("i_frame_num", "p_frame_num", "ave_i_frame_bpp", "ave_i_frame_psnr", "ave_p_frame_bpp", "ave_p_frame_psnr",
        "ave_all_frame_bpp", "ave_all_frame_psnr", "ave_all_frame_psnr_y")


def bd_published(r1, d1, r2, d2):
    r1, d1, r2, d2 = map(np.asarray, (r1, d1, r2, d2))
    l1, l2 = np.log(r1), np.log(r2)
    lo, hi = max(d1.min(), d2.min()), min(d1.max(), d2.max())
    xs, step = np.linspace(lo, hi, num=100, retstep=True)
    v1 = scipy.interpolate.pchip_interpolate(np.sort(d1), l1[np.argsort(d1)], xs)
    v2 = scipy.interpolate.pchip_interpolate(np.sort(d2), l2[np.argsort(d2)], xs)
    i1, i2 = np.trapz(v1, dx=step), np.trapz(v2, dx=step)
    return float((np.exp((i2 - i1) / (hi - lo)) - 1) * 100)


def main():
    with open(SRC) as f:
        full = json.load(f)
    anchor = {ds: {seq: {rp: {k: e[k] for k in KEEP} for rp, e in pts.items()} for seq, pts in full[ds].items()}
              for ds in ("UVG", "HEVC_B")}
    expected = {"split": {}, "cheaper": {}}
    for ds, seqs in anchor.items():
        for seq, pts in seqs.items():
            rps = sorted(pts)
            bpp = [pts[r]["ave_all_frame_bpp"] for r in rps]
            psnr = [pts[r]["ave_all_frame_psnr"] for r in rps]
            expected["split"][seq] = bd_published(bpp[1::2], psnr[1::2], bpp[0::2], psnr[0::2])
            expected["cheaper"][seq] = bd_published(bpp, psnr, [0.9 * b for b in bpp], psnr)
    with open(os.path.join(HERE, "bd_rate_golden.json"), "w") as f:
        json.dump({"anchor": anchor, "expected": expected}, f, indent=0, sort_keys=True)
    print({k: round(v, 4) for k, v in list(expected["split"].items())[:4]})


if __name__ == "__main__":
    main()
