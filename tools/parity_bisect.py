#!/usr/bin/env python
"""Where does the HIP path leave the oracle at full picture size? (TEST TOOL - runs on the GPU box.)

Runs the CPU oracle's compress() of one case of tests/golden/fullsize_digests.json and, for EVERY
dense operator the oracle evaluates (conv1x1 family, k x k conv, depthwise 3x3, transposed conv), runs
the SAME operator on the GPU through the kernel C ABI on the SAME inputs and compares the outputs
bit for bit. The oracle's own result is always carried forward, so every operator is tested on valid
inputs and the first report is the first operator that really differs. Prints, per mismatch, the
operator, the element, both values and - for contractions - the replayed accumulation
(oracle/nn_oracle.c orc_mfma16 step by step) of that output.

  python tools/parity_bisect.py dmci_1280x720_q32_t0.0 [max_reports]
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import nn  # noqa: E402

REPORTS = []
MAXREP = 6
STATS = {"ops": 0, "bad_ops": 0}


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def replay(x_row, w_row, bias):
    """accumulation of one output element, 16 products at a time (the oracle's routine)"""
    acc = np.float32(0.0 if bias is None else np.float32(bias))
    trail = [acc]
    for k in range(0, len(x_row), 16):
        acc = nn.mfma16(acc, w_row[k:k + 16], x_row[k:k + 16])
        trail.append(acc)
    return trail


def hardware_chain(x_row, w_row, trail):
    """every 16-block of the chain on the matrix core itself (tools/mfma_probe2.hip) with the oracle's
    accumulator as C: which block does the model get wrong?"""
    import subprocess
    exe = os.path.join(ROOT, "tools", "_bin", "mfma_probe2")
    if not os.path.exists(exe):
        return ["(tools/_bin/mfma_probe2 not built)"]
    T = len(x_row) // 16
    rec = np.zeros(T, dtype=[("a", np.float16, 16), ("b", np.float16, 16), ("c", np.float32)])
    for t in range(T):
        rec["a"][t] = w_row[16 * t:16 * t + 16]
        rec["b"][t] = x_row[16 * t:16 * t + 16]
        rec["c"][t] = trail[t]
    fin, fout = "/tmp/bisect_in.bin", "/tmp/bisect_out.bin"
    with open(fin, "wb") as f:
        f.write(np.int32(T).tobytes())
        f.write(rec.tobytes())
    subprocess.run([exe, fin, fout], check=True, capture_output=True)
    out = np.fromfile(fout, dtype=[("d", np.float32), ("bad", np.int32)])
    lines = []
    for t in range(T):
        if out["d"][t].tobytes() != np.float32(trail[t + 1]).tobytes():
            lines.append("block %d: C=%r hardware %r (0x%08x)  model %r (0x%08x)  a=%s b=%s" % (
                t, float(trail[t]), float(out["d"][t]), int(out["d"][t:t + 1].view(np.uint32)[0]), float(trail[t + 1]),
                int(np.array([trail[t + 1]], dtype=np.float32).view(np.uint32)[0]),
                rec["a"][t].view(np.uint16).tolist(), rec["b"][t].view(np.uint16).tolist()))
            MODEL_MISSES.append(dict(a=rec["a"][t].copy(), b=rec["b"][t].copy(), c=np.float32(trail[t]), d=out["d"][t]))
    return lines or ["every block of the chain: hardware == model (the difference is behind the contraction)"]


MODEL_MISSES = []


def report(name, got, want, detail=None):
    STATS["ops"] += 1
    bad = got.view(np.uint16) != want.view(np.uint16)
    n = int(bad.sum())
    if n == 0:
        return
    STATS["bad_ops"] += 1
    idx = np.argwhere(bad)
    if len(REPORTS) < MAXREP:
        lines = ["MISMATCH in %s: %d of %d outputs" % (name, n, want.size)]
        for i in idx[:4]:
            i = tuple(int(v) for v in i)
            lines.append("   at %s: gpu %r (0x%04x)  oracle %r (0x%04x)" % (
                i, float(got[i]), int(got.view(np.uint16)[i]), float(want[i]), int(want.view(np.uint16)[i])))
            if detail is not None:
                lines += ["      " + s for s in detail(i)]
        REPORTS.append("\n".join(lines))
        print(REPORTS[-1], flush=True)


def main():
    global MAXREP
    from gpu_util import Ops, call, ptr, stream
    from codec_util import dmci_model, oracle_for, picture
    name = sys.argv[1] if len(sys.argv) > 1 else "dmci_1280x720_q32_t0.0"
    if len(sys.argv) > 2:
        MAXREP = int(sys.argv[2])
    with open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")) as f:
        d = json.load(f)[name]
    assert d["kind"] == "dmci", "intra cases only"
    ops = Ops()
    m = dmci_model(skip_thres=d["skip_thres"])
    cdf = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_cdf.npz"))
    m.set_cdf_info(*[cdf["dmci_" + k].astype(np.int32) for k in ("z_cdf", "z_len", "y_cdf", "y_len")])
    o = oracle_for(m)
    x = picture(d["height"], d["width"], index=d["index"])

    orig_conv1x1, orig_dw, orig_kxk, orig_subpel = nn.conv1x1, nn.dwconv3x3, nn.conv_kxk, nn.subpel_conv1x1
    state = {"inner": False, "n": 0}

    def conv1x1(x, w, bias=None, r1=None, r2=None, q=None, q2=None, wsilu=False, chunk_add=False):
        want = orig_conv1x1(x, w, bias, r1=r1, r2=r2, q=q, q2=q2, wsilu=wsilu, chunk_add=chunk_add)
        if state["inner"]:
            return want
        K = x.shape[-1]
        x2 = np.ascontiguousarray(x.reshape(-1, K), dtype=np.float16)
        w2 = np.ascontiguousarray(w.reshape(w.shape[0], -1), dtype=np.float16)
        P, N = x2.shape[0], w2.shape[0]
        nout = N // 4 if chunk_add else N
        if K % 64 or N % 64:
            return want                                   # not a GPU contraction shape
        h = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float16)
        xd, wd, bd = dev(x2), dev(w2), dev(h(bias))
        r1d = dev(None if r1 is None else h(r1).reshape(P, nout))
        r2d = dev(None if r2 is None else h(r2).reshape(P, nout))
        qd, q2d = dev(h(q)), dev(h(q2))
        y = torch.zeros((P, nout), dtype=torch.half, device="cuda")
        call(ops.conv1x1, ptr(xd), K, ptr(wd), ptr(bd), ptr(r1d), nout, ptr(r2d), nout, ptr(qd), ptr(q2d),
             ptr(y), nout, P, K, N, (1 if wsilu else 0) | (2 if chunk_add else 0), stream())
        torch.cuda.synchronize()
        state["n"] += 1

        def detail(i):
            p_, n_ = i[0] if len(i) == 2 else np.ravel_multi_index(i[:-1], want.shape[:-1]), i[-1]
            out = []
            chans = range(4 * n_, 4 * n_ + 4) if chunk_add else [n_]
            for c in chans:
                tr = replay(x2[p_], w2[c], None if bias is None else h(bias)[c])
                out.append("channel %d: bias %r, accumulator after each 16-block: %s" % (
                    c, None if bias is None else float(h(bias)[c]), " ".join("%.9g" % float(t) for t in tr[-4:])))
                out += hardware_chain(x2[p_], w2[c], tr)
            return out
        report("op #%d conv1x1 P=%d K=%d N=%d%s%s%s%s%s" % (
            state["n"], P, K, N, " wsilu" if wsilu else "", " chunk_add" if chunk_add else "",
            " r1" if r1 is not None else "", " r2" if r2 is not None else "",
            " q" if q is not None else "" + (" q2" if q2 is not None else "")),
            y.cpu().numpy().reshape(want.shape), want, detail)
        return want

    def dwconv3x3(x, w):
        want = orig_dw(x, w)
        H, W, C = x.shape
        wt = np.ascontiguousarray(np.transpose(w[:, 0], (1, 2, 0)).reshape(9, C), dtype=np.float16)
        xd, wd = dev(np.ascontiguousarray(x, dtype=np.float16)), dev(wt)
        y = torch.zeros((H, W, C), dtype=torch.half, device="cuda")
        call(ops.dwconv3x3, ptr(xd), C, ptr(wd), ptr(y), C, H, W, C, stream())
        torch.cuda.synchronize()
        state["n"] += 1
        report("op #%d dwconv3x3 %dx%dx%d" % (state["n"], H, W, C), y.cpu().numpy(), want)
        return want

    def conv_kxk(x, w, bias, ksize, stride, pad):
        state["inner"] = True
        try:
            want = orig_kxk(x, w, bias, ksize, stride, pad)
        finally:
            state["inner"] = False
        H, W, C = x.shape
        cout = w.shape[0]
        if C % 64 or cout % 64:
            return want
        wt = np.ascontiguousarray(np.transpose(w, (0, 2, 3, 1)), dtype=np.float16)
        xd, wd, bd = dev(np.ascontiguousarray(x, dtype=np.float16)), dev(wt), dev(np.ascontiguousarray(bias, dtype=np.float16))
        y = torch.zeros(want.shape, dtype=torch.half, device="cuda")
        call(ops.conv_kxk, ptr(xd), C, ptr(wd), ptr(bd), ptr(y), cout, H, W, C, cout, ksize, stride, pad, stream())
        torch.cuda.synchronize()
        state["n"] += 1
        xh = np.ascontiguousarray(x, dtype=np.float16)

        def detail(i):
            ho, wo, c = i
            row = np.zeros((ksize, ksize, C), dtype=np.float16)       # contraction index (ky, kx, cin)
            for ky in range(ksize):
                for kx in range(ksize):
                    hi, wi = ho * stride - pad + ky, wo * stride - pad + kx
                    if 0 <= hi < H and 0 <= wi < W:
                        row[ky, kx] = xh[hi, wi]
            x_row, w_row = row.reshape(-1), wt[c].reshape(-1)
            tr = replay(x_row, w_row, np.ascontiguousarray(bias, dtype=np.float16)[c])
            out = ["channel %d: accumulator after the last 16-blocks: %s" % (c, " ".join("%.9g" % float(t) for t in tr[-4:]))]
            return out + hardware_chain(x_row, w_row, tr)
        report("op #%d conv %dx%d s%d %dx%dx%d -> %d" % (state["n"], ksize, ksize, stride, H, W, C, cout),
               y.cpu().numpy(), want, detail)
        return want

    def subpel_conv1x1(x, w):
        state["inner"] = True
        try:
            want = orig_subpel(x, w)
        finally:
            state["inner"] = False
        H, W, C = x.shape
        cout = w.shape[0] // 4
        wt = np.ascontiguousarray(w.reshape(cout, 4, C).transpose(1, 0, 2), dtype=np.float16)     # [dy*2+dx][cout][cin]
        xd, wd = dev(np.ascontiguousarray(x, dtype=np.float16)), dev(wt)
        y = torch.zeros(want.shape, dtype=torch.half, device="cuda")
        call(ops.tconv2x2, ptr(xd), C, ptr(wd), ptr(y), cout, H, W, C, cout, stream())
        torch.cuda.synchronize()
        state["n"] += 1
        report("op #%d tconv2x2 %dx%dx%d -> %d" % (state["n"], H, W, C, cout), y.cpu().numpy(), want)
        return want

    nn.conv1x1, nn.dwconv3x3, nn.conv_kxk, nn.subpel_conv1x1 = conv1x1, dwconv3x3, conv_kxk, subpel_conv1x1
    r = o.compress(x, d["qp"])
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes() if not isinstance(a, bytes) else a).hexdigest()
    print("oracle digests reproduce here: y %s, z %s, bytes %s" % (
        sha(o.debug["y"]) == d["y"], sha(o.debug["z_i8"]) == d["z_i8"], sha(r["bit_stream"]) == d["bit_stream"]))
    print("%d dense operators compared with the GPU on identical inputs, %d differ" % (STATS["ops"], STATS["bad_ops"]))
    if MODEL_MISSES:
        out = os.path.join(ROOT, "gpurun_out", "mfma_model_misses_%s.npz" % name)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        np.savez(out, a=np.stack([m["a"] for m in MODEL_MISSES]), b=np.stack([m["b"] for m in MODEL_MISSES]),
                 c=np.array([m["c"] for m in MODEL_MISSES], dtype=np.float32),
                 d=np.array([m["d"] for m in MODEL_MISSES], dtype=np.float32))
        print("matrix-core trials the model gets wrong saved to", out)


if __name__ == "__main__":
    main()
