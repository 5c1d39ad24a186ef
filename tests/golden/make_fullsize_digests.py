#!/usr/bin/env python
"""Oracle-derived digests at BASELINE's own picture sizes (offline, CPU box).

The codec-level HIP-vs-oracle tests run the oracle live at <= 128x96 (seconds). At 720p / 1080p the
oracle takes minutes per picture, so it runs HERE, once, and the sha256 of every tensor the small
tests compare ({bit stream, y, z_i8, y_hat, x_hat, temporal state}) is committed as
tests/golden/fullsize_digests.json; tests/test_fullsize_gpu.py compares the HIP codecs with those
digests on the GPU box (where neither /root/reference nor minutes of oracle time are available).

Inputs are the seeded synthetic pictures / weights of dcvc_amd/synthetic.py; their own sha256 is
stored too, so a host whose numpy / torch builds produce different synthetic data is detected as
such instead of being reported as a codec mismatch.

  python tests/golden/make_fullsize_digests.py [--only NAME_SUBSTRING] [--quick]

Cases: DMCI 256x256 (BASELINE configs[0] tile), 1280x720 and 1920x1080 with
q in {0, 32, 63} x skip_thres in {0.15, 0}; LD 1080p I + 3 P with a reset; HT-S one chunk at 720p and
1080p; HT-L one chunk at 1080p. Round 3: DMCI 1080p q 16 / 48, HT-L 720p, DMCI and
one LD P picture at 3840x2160. Existing entries are kept unless --force. All digests depend on the arithmetic
policy: round 3's policy v3 (WSiLU) regenerated every one of them.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OUT = os.path.join(HERE, "fullsize_digests.json")


def sha(a):
    if isinstance(a, (bytes, bytearray)):
        return hashlib.sha256(bytes(a)).hexdigest()
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def state_dict_digest(model):
    h = hashlib.sha256()
    sd = model.state_dict()
    for k in sorted(sd):
        t = sd[k]
        a = t.detach().cpu().numpy()
        if a.dtype.kind == "f":
            a = a.astype(np.float16)
        h.update(k.encode())
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


CDF_OUT = os.path.join(HERE, "fullsize_cdf.npz")


def save_cdf_tables():
    """The entropy tables the digests were made with (update() is host-dependent in the last place:
    an Intel and an AMD host quantise a few frequencies differently); int32 -> uint16/uint8, compressed."""
    from codec_util import dmc_ht_model, dmc_ld_model, dmci_model
    out = {}
    for kind, m in (("dmci", dmci_model(skip_thres=0.15)), ("ld", dmc_ld_model(skip_thres=0.15)),
                    ("hts", dmc_ht_model("hts", skip_thres=0.15)), ("htl", dmc_ht_model("htl", skip_thres=0.15))):
        z_cdf, z_len, y_cdf, y_len = [np.asarray(t) for t in m.get_cdf_info()]
        assert z_cdf.max() <= 65536 and z_len.max() < 256
        out[kind + "_z_cdf"] = z_cdf.astype(np.uint32)
        out[kind + "_z_len"] = z_len.astype(np.uint8)
        out[kind + "_y_cdf"] = y_cdf.astype(np.uint32)
        out[kind + "_y_len"] = y_len.astype(np.uint8)
    np.savez_compressed(CDF_OUT, **out)


def padded(x):
    h, w, _ = x.shape
    return np.pad(x, ((0, -h % 16), (0, -w % 16), (0, 0)), mode="edge")


def run_dmci(hw, qp, thres, decode):
    from codec_util import dmci_model, oracle_for, picture
    m = dmci_model(skip_thres=thres)
    o = oracle_for(m)
    x = picture(hw[0], hw[1], index=qp)
    t = time.time()
    r = o.compress(x, qp)
    d = dict(kind="dmci", height=hw[0], width=hw[1], qp=qp, skip_thres=thres, index=qp,
             weights=state_dict_digest(m), input=sha(x),
             bytes=len(r["bit_stream"]), ec_parallel=int(r["ec_parallel"]),
             bit_stream=sha(r["bit_stream"]), y=sha(o.debug["y"]), z_i8=sha(o.debug["z_i8"]),
             y_hat=sha(o.debug["y_hat"]), x_hat=sha(r["x_hat"]))
    if decode:
        xd = o.decompress(r["bit_stream"], qp, hw[0], hw[1], r["ec_parallel"])
        assert np.array_equal(xd, r["x_hat"]), "oracle closure"
        d["oracle_closure_checked"] = True
    d["psnr"] = float(psnr(r["x_hat"][:hw[0], :hw[1]], x))
    d["oracle_seconds"] = round(time.time() - t, 1)
    return d


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)


def run_inter(kind, hw, plan, thres):
    from codec_util import chunk, dmc_ht_model, dmc_ld_model, oracle_for, picture
    if kind == "ld":
        m = dmc_ld_model(skip_thres=thres)
        frames = 1
    else:
        m = dmc_ht_model(kind, skip_thres=thres)
        frames = 8
    enc, dec = oracle_for(m), oracle_for(m)
    ref = padded(picture(hw[0], hw[1], index=0))
    t = time.time()
    enc.add_ref_feature_from_frame(ref, True)
    dec.add_ref_feature_from_frame(ref, False)
    d = dict(kind=kind, height=hw[0], width=hw[1], skip_thres=thres, plan=[list(p) for p in plan],
             weights=state_dict_digest(m), ref=sha(ref), calls=[])
    d["state0"] = {"memory": sha(enc.memory), "ctx": sha(enc.ctx)}
    for i, (qp, reset) in enumerate(plan):
        x = picture(hw[0], hw[1], index=i + 1) if frames == 1 else chunk(hw[0], hw[1], 1 + 8 * i)
        r = enc.compress(x, qp, bool(reset))
        xd = dec.decompress(r["bit_stream"], qp, hw[0], hw[1], r["ec_parallel"], bool(reset))
        if isinstance(xd, (list, tuple)):
            xd = np.concatenate(xd, axis=-1)
        assert np.array_equal(dec.feature_p, enc.feature_p), "oracle encoder / decoder lock-step"
        d["calls"].append(dict(
            qp=qp, reset=int(reset), input=sha(x), bytes=len(r["bit_stream"]), ec_parallel=int(r["ec_parallel"]),
            bit_stream=sha(r["bit_stream"]), y=sha(enc.debug["y"]), z_i8=sha(enc.debug["z_i8"]),
            y_hat=sha(enc.debug["y_hat"]), feature_p=sha(enc.feature_p), memory=sha(enc.memory),
            ctx=sha(enc.ctx), x_hat=sha(xd), psnr=float(psnr(xd[:hw[0], :hw[1]], x))))
        print("   call", i, "bytes", len(r["bit_stream"]), "psnr %.2f" % d["calls"][-1]["psnr"], flush=True)
    d["oracle_seconds"] = round(time.time() - t, 1)
    return d


def cases(quick):
    c = {}
    c["dmci_256x256_q32_t0.15"] = lambda: run_dmci((256, 256), 32, 0.15, True)
    sizes = [(720, 1280)] if quick else [(720, 1280), (1080, 1920)]
    for hw in sizes:
        for qp in (0, 32, 63):
            for thres in (0.15, 0.0):
                name = "dmci_%dx%d_q%d_t%s" % (hw[1], hw[0], qp, thres)
                c[name] = (lambda hw=hw, qp=qp, thres=thres: run_dmci(hw, qp, thres, decode=(qp == 32)))
    if not quick:
        # round 3: the two remaining rate points bench.py cycles through (q 16, 48), an HT-L case below 1080p, and
        # 3840x2160 (BASELINE configs[4]; README.md:202 lists it as a tuned size) - intra and one LD P picture
        for qp in (16, 48):
            c["dmci_1920x1080_q%d_t0.15" % qp] = (lambda qp=qp: run_dmci((1080, 1920), qp, 0.15, decode=False))
        c["htl_1280x720"] = lambda: run_inter("htl", (720, 1280), [(32, 0)], 0.15)
        c["dmci_3840x2160_q32_t0.15"] = lambda: run_dmci((2160, 3840), 32, 0.15, decode=False)
        c["ld_3840x2160"] = lambda: run_inter("ld", (2160, 3840), [(32, 0)], 0.15)
        c["ld_1920x1080"] = lambda: run_inter("ld", (1080, 1920), [(32, 0), (40, 1), (40, 0)], 0.15)
        c["hts_1920x1080"] = lambda: run_inter("hts", (1080, 1920), [(32, 0)], 0.15)
        c["htl_1920x1080"] = lambda: run_inter("htl", (1080, 1920), [(32, 0)], 0.15)
    c["ld_1280x720"] = lambda: run_inter("ld", (720, 1280), [(32, 0), (40, 1), (40, 0)], 0.15)
    if not quick:
        # round 4: the rate points at both ends of the range for the inter models (the full-size
        # cases above code at q 32 / 40 / 45 only, bench.py cycles 0 ... 63), a second HT-L call WITH a memory reset,
        # and the hierarchical models at 3840x2160 (BASELINE configs[4])
        c["ld_qends_1280x720"] = lambda: run_inter("ld", (720, 1280), [(0, 0), (63, 0), (0, 1)], 0.15)
        c["hts_qends_1280x720"] = lambda: run_inter("hts", (720, 1280), [(0, 0), (63, 0)], 0.15)
        c["htl_qends_1280x720"] = lambda: run_inter("htl", (720, 1280), [(0, 0), (63, 1)], 0.15)
        # ... and the two remaining rate points bench.py cycles through
        c["ld_qmid_1280x720"] = lambda: run_inter("ld", (720, 1280), [(16, 0), (48, 0)], 0.15)
        c["hts_qmid_1280x720"] = lambda: run_inter("hts", (720, 1280), [(16, 0), (48, 0)], 0.15)
        c["htl_qmid_1280x720"] = lambda: run_inter("htl", (720, 1280), [(16, 0), (48, 0)], 0.15)
        c["hts_3840x2160"] = lambda: run_inter("hts", (2160, 3840), [(32, 0)], 0.15)
        c["htl_3840x2160"] = lambda: run_inter("htl", (2160, 3840), [(32, 0)], 0.15)
    c["hts_1280x720"] = lambda: run_inter("hts", (720, 1280), [(32, 0), (45, 1)], 0.15)
    if not quick:
        # round 5: where the coverage was thinnest - every inter model at 1920x1080 with the rate
        # points at both ends of the range AND a memory reset, the intra model at 3840x2160 at q 0 and q 63, LD at
        # 3840x2160 at q 0 / 63 with a reset
        c["ld_qends_1920x1080"] = lambda: run_inter("ld", (1080, 1920), [(0, 0), (63, 1), (63, 0)], 0.15)
        c["dmci_3840x2160_q63_t0.15"] = lambda: run_dmci((2160, 3840), 63, 0.15, decode=False)
        c["hts_qends_1920x1080"] = lambda: run_inter("hts", (1080, 1920), [(0, 0), (63, 1)], 0.15)
        c["dmci_3840x2160_q0_t0.15"] = lambda: run_dmci((2160, 3840), 0, 0.15, decode=False)
        c["htl_qends_1920x1080"] = lambda: run_inter("htl", (1080, 1920), [(63, 0), (0, 1)], 0.15)
        c["ld_qends_3840x2160"] = lambda: run_inter("ld", (2160, 3840), [(63, 0), (0, 1)], 0.15)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    save_cdf_tables()
    done = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            done = json.load(f)
    for name, fn in cases(a.quick).items():
        if a.only and a.only not in name:
            continue
        if name in done and not a.force:
            continue
        print("==", name, flush=True)
        done[name] = fn()
        print("   %.0f s" % done[name]["oracle_seconds"], flush=True)
        with open(OUT + ".tmp", "w") as f:
            json.dump(done, f, indent=1, sort_keys=True)
        os.replace(OUT + ".tmp", OUT)


if __name__ == "__main__":
    main()
