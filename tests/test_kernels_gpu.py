"""Per-kernel parity on a real MI355X (-m gpu), through the C ABI.

Dense convolutions are checked against a plain PyTorch fp32 reference of the same op (fp16
inputs, fp32 math, one rounding) within a 1-2 fp16-ulp tolerance; integer / layout / symbol
kernels are checked bit-exactly against the numpy oracle (oracle/symbols_np.py)."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs the MI355X"
    from gpu_util import Ops
    return Ops()


def _rand(shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half()


def _wsilu(x):
    return x * torch.sigmoid(4.0 * x)


def _close(got, want, what, rtol=2e-3, atol=2e-3):
    got = got.float().cpu()
    want = want.float().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = (err > tol).sum().item()
    print("%s: max abs err %.3e, max |want| %.3e, violations %d / %d" %
          (what, err.max().item(), want.abs().max().item(), bad, err.numel()))
    assert bad == 0, what


CONV_CASES = [
    # pixels, cin, cout, ldx_extra, flags/residual config
    dict(P=1000, K=192, N=384, name="bias"),
    dict(P=384, K=384, N=384, wsilu=True, name="bias_wsilu"),
    dict(P=777, K=256, N=512, r1=True, name="bias_shortcut"),
    dict(P=640, K=128, N=128, r1=True, r2=True, name="bias_shortcut2"),
    dict(P=500, K=128, N=256, r1=True, q=True, name="bias_shortcut_with_quant"),
    dict(P=300, K=256, N=256, q=True, name="bias_with_quant"),
    dict(P=900, K=128, N=512, wsilu=True, chunk=True, name="bias_wsilu_chunk_add"),
    dict(P=2048, K=384, N=1536, wsilu=True, chunk=True, name="bias_wsilu_chunk_add_384"),
    dict(P=129, K=512, N=192, q2=True, name="bias_then_scale_n192"),
    dict(P=4096, K=2048, N=512, name="bias_k2048"),
    dict(P=5000, K=1024, N=128, r1=True, q=True, name="narrow_small_grid_64x64"),
    dict(P=777, K=512, N=64, wsilu=True, name="narrow_small_grid_64x64_wsilu"),
    # 1080p-sized grids: the 256-pixel tile shapes (256x256 and 256x192), ragged last tile
    dict(P=32641, K=384, N=1536, wsilu=True, chunk=True, name="p8_chunk_add_256x256"),
    dict(P=32600, K=384, N=384, r1=True, name="p8_shortcut_256x192"),
    dict(P=32640, K=192, N=768, wsilu=True, name="p8_wsilu_256x256"),
    dict(P=30000, K=384, N=384, r1=True, r2=True, q2=True, name="p8_shortcut2_scale_256x192"),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c["name"] for c in CONV_CASES])
def test_conv1x1(ops, case):
    from gpu_util import call, ptr, stream
    from dcvc_amd.plugin import MLCodec_extensions_cpp  # noqa: F401  (loads the library)
    P, K, N = case["P"], case["K"], case["N"]
    dev = "cuda"
    ldx = K + 64                      # activations are a channel slice of a wider buffer
    xw = _rand((P, ldx), 1.0, 1).to(dev)
    x = xw[:, 32:32 + K]
    w = _rand((N, K), 1.0 / K ** 0.5, 2).to(dev)
    b = _rand((N,), 0.5, 3).to(dev)
    nout = N // 4 if case.get("chunk") else N
    r1 = _rand((P, nout), 1.0, 4).to(dev) if case.get("r1") else None
    r2 = _rand((P, nout), 1.0, 5).to(dev) if case.get("r2") else None
    q = (_rand((nout,), 0.3, 6) + 1).to(dev) if case.get("q") else None
    q2 = (_rand((nout,), 0.3, 7) + 1).to(dev) if case.get("q2") else None
    ldy = nout + 8
    yw = torch.zeros((P, ldy), dtype=torch.half, device=dev)
    flags = (1 if case.get("wsilu") else 0) | (2 if case.get("chunk") else 0)
    call(ops.conv1x1, ptr(x), ldx, ptr(w), ptr(b), ptr(r1), nout, ptr(r2), nout, ptr(q), ptr(q2),
         ptr(yw), ldy, P, K, N, flags, stream())
    torch.cuda.synchronize()
    acc = x.float() @ w.float().t() + b.float()
    if case.get("wsilu"):
        acc = _wsilu(acc)
    if case.get("chunk"):
        acc = acc.view(P, N // 4, 4).sum(-1)
    if r1 is not None:
        acc = acc + r1.float()
    if r2 is not None:
        acc = acc + r2.float()
    if q is not None:
        acc = acc * q.float()
    want = acc.half()
    if q2 is not None:
        want = (want.float() * q2.float()).half()
    _close(yw[:, :nout], want, "conv1x1 " + case["name"])
    assert torch.count_nonzero(yw[:, nout:]).item() == 0, "wrote outside the channel slice"
    # bit-exact against the oracle (measured matrix-core arithmetic, oracle/nn_oracle.c)
    from oracle import nn
    np_ = lambda t: None if t is None else t.cpu().numpy()
    rows = slice(max(0, P - 400), P) if P > 20000 else slice(0, min(P, 640))
    orc = nn.conv1x1(np_(x)[rows], np_(w), np_(b), r1=None if r1 is None else np_(r1)[rows],
                     r2=None if r2 is None else np_(r2)[rows], q=np_(q), q2=np_(q2),
                     wsilu=bool(case.get("wsilu")), chunk_add=bool(case.get("chunk")))
    got = yw[rows, :nout].cpu().numpy()
    bad = int((got != orc).sum())
    print("   vs oracle: %d mismatching of %d" % (bad, got.size))
    assert bad == 0


@pytest.mark.parametrize("k,s,p,H,W,cin,cout", [(3, 2, 1, 18, 30, 128, 128), (2, 2, 0, 34, 60, 64, 128),
                                                (3, 1, 1, 17, 15, 64, 256), (3, 2, 1, 17, 31, 192, 256),
                                                # round 6: 64 x 64 tiles for narrow layers on small grids (LD's encoder.down at 1080p)
                                                (3, 2, 1, 136, 240, 256, 128), (3, 2, 1, 70, 100, 256, 128)])
def test_conv_kxk(ops, k, s, p, H, W, cin, cout):
    from gpu_util import call, ptr, stream, nhwc
    dev = "cuda"
    x = _rand((1, cin, H, W), 1.0, 11).to(dev)
    w = _rand((cout, cin, k, k), 1.0 / (cin * k * k) ** 0.5, 12).to(dev)
    b = _rand((cout,), 0.5, 13).to(dev)
    want = F.conv2d(x.float(), w.float(), b.float(), stride=s, padding=p).half()
    Ho, Wo = want.shape[2], want.shape[3]
    xh = nhwc(x)
    wt = w.permute(0, 2, 3, 1).contiguous()          # [cout][ky][kx][cin]
    y = torch.zeros((Ho, Wo, cout), dtype=torch.half, device=dev)
    call(ops.conv_kxk, ptr(xh), cin, ptr(wt), ptr(b), ptr(y), cout, H, W, cin, cout, k, s, p, stream())
    torch.cuda.synchronize()
    _close(y, nhwc(want), "conv %dx%d s%d" % (k, k, s))
    from oracle import nn
    orc = nn.conv_kxk(xh.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), k, s, p)
    assert np.array_equal(y.cpu().numpy(), orc)


def test_tconv2x2(ops):
    from gpu_util import call, ptr, stream, nhwc
    dev = "cuda"
    H, W, cin, cout = 17, 30, 256, 384
    x = _rand((1, cin, H, W), 1.0, 21).to(dev)
    w = _rand((cout * 4, cin, 1, 1), 1.0 / cin ** 0.5, 22).to(dev)
    want = F.pixel_shuffle(F.conv2d(x.float(), w.float()), 2).half()      # SubpelConv2x, layers.py:92-103
    wt = w[:, :, 0, 0].view(cout, 4, cin).permute(1, 0, 2).contiguous()   # [dy*2+dx][cout][cin]
    y = torch.zeros((2 * H, 2 * W, cout), dtype=torch.half, device=dev)
    call(ops.tconv2x2, ptr(nhwc(x)), cin, ptr(wt), ptr(y), cout, H, W, cin, cout, stream())
    torch.cuda.synchronize()
    _close(y, nhwc(want), "tconv2x2")
    from oracle import nn
    assert np.array_equal(y.cpu().numpy(), nn.subpel_conv1x1(nhwc(x).cpu().numpy(), w.cpu().numpy()))


@pytest.mark.parametrize("H,W,C", [(17, 30, 384), (5, 7, 64), (68, 120, 256), (136, 240, 384), (13, 9, 128), (6, 3, 8), (8, 2, 64),
                                   (1, 1, 8), (7, 64, 16), (45, 80, 128)])
def test_dwconv3x3(ops, H, W, C):
    from gpu_util import call, ptr, stream, nhwc
    dev = "cuda"
    x = _rand((1, C, H, W), 1.0, 31).to(dev)
    w = _rand((C, 1, 3, 3), 0.3, 32).to(dev)
    want = F.conv2d(x.float(), w.float(), None, padding=1, groups=C).half()
    wt = w[:, 0].permute(1, 2, 0).reshape(9, C).contiguous()
    y = torch.zeros((H, W, C), dtype=torch.half, device=dev)
    call(ops.dwconv3x3, ptr(nhwc(x)), C, ptr(wt), ptr(y), C, H, W, C, stream())
    torch.cuda.synchronize()
    _close(y, nhwc(want), "dwconv3x3", rtol=1e-3, atol=1e-3)
    from oracle import nn
    assert np.array_equal(y.cpu().numpy(), nn.dwconv3x3(nhwc(x).cpu().numpy(), w.cpu().numpy()))
    # strided input / output rows (the codecs pass channel-slice views) give the same numbers
    xs = torch.zeros((H, W, C + 24), dtype=torch.half, device=dev)
    xs[:, :, 8:8 + C] = nhwc(x)
    ys = torch.zeros((H, W, C + 40), dtype=torch.half, device=dev)
    call(ops.dwconv3x3, ctypes.c_void_p(xs.data_ptr() + 16), C + 24, ptr(wt), ctypes.c_void_p(ys.data_ptr() + 32), C + 40, H, W, C, stream())
    torch.cuda.synchronize()
    assert torch.equal(ys[:, :, 16:16 + C], y) and float(ys[:, :, :16].abs().max()) == 0 and float(ys[:, :, 16 + C:].abs().max()) == 0


@pytest.mark.parametrize("variant", ["8,1", "8,2", "8,3", "4,1", "4,2", "16,1", "16,2"])      # every selectable instantiation
def test_dwconv3x3_other_variants(variant):
    """the other instantiations of the depthwise walk (rows per lane, rows of loads in flight; DCVC_DWCONV_VARIANT, read once
    per process): same arithmetic, same bits; prints the launch time of each at 1080p / 8"""
    import subprocess
    import sys
    env = dict(os.environ, DCVC_DWCONV_VARIANT=variant)
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-s", "-k",
                          "test_dwconv3x3 and not other_variants", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600)
    print("\n".join(l for l in res.stdout.splitlines() if "us per launch" in l))
    import re
    m = re.search(r"(\d+) passed", res.stdout)
    assert res.returncode == 0 and m and int(m.group(1)) >= 10 and "failed" not in res.stdout, res.stdout[-3000:] + res.stderr[-1000:]


def test_dwconv3x3_unknown_variant_is_refused():
    """a typo in DCVC_DWCONV_VARIANT used to select round 1's kernel silently (advisor, round 5): now an error"""
    import subprocess
    import sys
    code = ("import torch, sys; sys.path.insert(0, %r); import gpu_util as g; ops = g.Ops(); "
            "x = torch.zeros((8, 8, 64), dtype=torch.half, device='cuda'); w = torch.zeros((9, 64), dtype=torch.half, device='cuda'); "
            "g.call(ops.dwconv3x3, g.ptr(x), 64, g.ptr(w), g.ptr(x.clone()), 64, 8, 8, 64, g.stream())" % os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DCVC_DWCONV_VARIANT="8,4"), capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "DCVC_DWCONV_VARIANT" in (res.stderr + res.stdout), res.stderr[-1500:]


def test_dwconv3x3_launch_time(ops):
    """not an assertion on speed - a record: microseconds per launch at (136 x 240, 384 channels) = 25 MB in, 25 MB out"""
    from gpu_util import call, ptr, stream
    dev = "cuda"
    H, W, C = 136, 240, 384
    x = _rand((H, W, C), 1.0, 33).to(dev)
    wt = _rand((9, C), 0.3, 34).to(dev)
    y = torch.zeros((H, W, C), dtype=torch.half, device=dev)
    for _ in range(5):
        call(ops.dwconv3x3, ptr(x), C, ptr(wt), ptr(y), C, H, W, C, stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        call(ops.dwconv3x3, ptr(x), C, ptr(wt), ptr(y), C, H, W, C, stream())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print("dwconv3x3 %s: %.2f us per launch, %.2f TB/s of the 50.1 MB a launch has to move" % (
        os.environ.get("DCVC_DWCONV_VARIANT", "default (8,2; 4,2 for <= 128 channels)"), us, 2 * H * W * C * 2 / us / 1e6))


def test_layout_kernels_exact(ops):
    from gpu_util import call, ptr, stream, nhwc
    dev = "cuda"
    H, W = 70, 100                       # pads to 80 x 112
    x = _rand((1, 3, H, W), 0.3, 41).to(dev)
    pb, pr = 80 - H, 112 - W
    xp = F.pad(x, (0, pr, 0, pb), mode="replicate")
    want = nhwc(F.pixel_unshuffle(xp, 8))
    out = torch.zeros((10, 14, 192), dtype=torch.half, device=dev)
    call(ops.pad_unshuffle8, ptr(nhwc(x)), H, W, 3, ptr(out), 10, 14, stream())
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    # shuffle8 + clamp is the inverse (on the clamped values)
    back = torch.zeros((80, 112, 3), dtype=torch.half, device=dev)
    call(ops.shuffle8, ptr(out), 192, 10, 14, 3, 1, ptr(back), stream())
    torch.cuda.synchronize()
    assert torch.equal(back, nhwc(xp.clamp(-0.5, 0.5)))
    # shuffle2
    t = _rand((1, 64, 6, 5), 1.0, 42).to(dev)
    s2 = torch.zeros((12, 10, 16), dtype=torch.half, device=dev)
    call(ops.shuffle2, ptr(nhwc(t)), 64, 6, 5, 16, ptr(s2), 16, stream())
    torch.cuda.synchronize()
    assert torch.equal(s2, nhwc(F.pixel_shuffle(t, 2)))
    # replicate pad / crop / mul_channel
    t = _rand((1, 32, 9, 7), 1.0, 43).to(dev)
    tp = torch.zeros((12, 8, 32), dtype=torch.half, device=dev)
    call(ops.replicate_pad, ptr(nhwc(t)), 32, 9, 7, 32, 3, 1, ptr(tp), 32, stream())
    cr = torch.zeros((9, 7, 32), dtype=torch.half, device=dev)
    call(ops.crop, ptr(tp), 32, 8, ptr(cr), 32, 9, 7, 32, stream())
    q = (_rand((32,), 0.3, 44) + 1).to(dev)
    mc = torch.zeros((9, 7, 32), dtype=torch.half, device=dev)
    call(ops.mul_channel, ptr(cr), 32, ptr(q), ptr(mc), 32, 63, 32, stream())
    torch.cuda.synchronize()
    assert torch.equal(tp, nhwc(F.pad(t, (0, 1, 0, 3), mode="replicate")))
    assert torch.equal(cr, nhwc(t))
    assert torch.equal(mc, (nhwc(t) * q))


def test_round_z_exact(ops):
    from gpu_util import call, ptr, stream
    from oracle import symbols_np as orc
    dev = "cuda"
    z = torch.cat([_rand((5000,), 30.0, 51), torch.tensor([0.5, -0.5, 1.5, 2.5, -2.5, 63.5, -64.5, 200, -200]).half()])
    zh = torch.zeros_like(z).to(dev)
    zi = torch.zeros(z.numel(), dtype=torch.int8, device=dev)
    call(ops.round_z, ptr(z.to(dev)), ptr(zh), ptr(zi), z.numel(), stream())
    torch.cuda.synchronize()
    wh, wi = orc.round_z(z.numpy())
    assert np.array_equal(zh.cpu().numpy(), wh)
    assert np.array_equal(zi.cpu().numpy(), wi)


@pytest.mark.parametrize("H,W,thres", [(17, 30, 0.15), (8, 8, 0.0), (68, 120, 0.15), (5, 3, 0.15)])
def test_y_steps_exact(ops, H, W, thres):
    """4-step masked quantisation + index + compaction (encoder), index + compaction (decoder) and
    the restore, all bit-exact against the op-by-op numpy restatement of the reference kernels."""
    from gpu_util import call, ptr, stream
    from oracle import symbols_np as orc
    dev = "cuda"
    C = 256
    cq = C // 4
    y = _rand((H, W, C), 6.0, 61)
    # scales: mix of negative, tiny, in-range and huge values; every fp16 bucket of the table
    sc = (_rand((H, W, C), 1.0, 62).float().exp() * 0.3).half()
    sc[0, 0, :8] = torch.tensor([0.0, -1.0, 0.11, 0.1099, 16.0, 17.0, 0.15, 0.1501]).half()
    mean_list = [_rand((H, W, C), 2.0, 63 + i) for i in range(4)]
    n = H * W * cq
    nb = ops.symbol_blocks(n)
    yd, scd = y.to(dev), sc.to(dev)
    acc = torch.full((H, W, C), 7.0, dtype=torch.half, device=dev)
    sym = torch.zeros(n, dtype=torch.int16, device=dev)
    cond = torch.zeros((n + 7) // 8, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(nb, dtype=torch.int32, device=dev)
    comp = torch.zeros(4 * n, dtype=torch.int16, device=dev)
    totals = torch.zeros(4, dtype=torch.int32, device=dev)
    masks = orc.get_mask_4x(H, W, C)
    want_acc = np.zeros((H, W, C), np.float16)
    want_syms = []
    keeps = []
    for step in range(4):
        md = mean_list[step].to(dev)
        call(ops.y_step_enc, ptr(yd), C, ptr(scd), C, ptr(md), C, ptr(acc), C, ptr(sym), ptr(cond),
             ptr(cnt), ptr(comp), ptr(totals), H, W, C, step, thres, stream())
        torch.cuda.synchronize()
        y_q, y_hat, s_hat = orc.process_with_mask(y.numpy(), sc.numpy(), mean_list[step].numpy(),
                                                  masks[step], thres)
        comb, keep = orc.build_index_enc(orc.fold4(y_q), orc.fold4(s_hat), thres)
        want_acc = (want_acc + y_hat).astype(np.float16)
        assert np.array_equal(sym.cpu().numpy(), comb), "symbols step %d" % step
        want_syms.append(comb[keep])
        keeps.append(keep)
        assert np.array_equal(acc.cpu().numpy(), want_acc), "y_hat_so_far step %d" % step
    tot = totals.cpu().numpy()
    assert list(tot) == [len(s) for s in want_syms]
    assert np.array_equal(comp.cpu().numpy()[:tot.sum()], np.concatenate(want_syms))

    # decoder: index + compaction, then restore from the "decoded" symbols
    idx = torch.zeros(n, dtype=torch.uint8, device=dev)
    cidx = torch.zeros(4 * n, dtype=torch.uint8, device=dev)
    totals_d = torch.zeros(4, dtype=torch.int32, device=dev)
    acc_d = torch.full((H, W, C), -3.0, dtype=torch.half, device=dev)
    decoded = torch.zeros(4 * n, dtype=torch.int8, device=dev)
    base = 0
    for step in range(4):
        call(ops.y_step_dec_index, ptr(scd), C, ptr(idx), ptr(cond), ptr(cnt), ptr(cidx),
             ptr(totals_d), H, W, C, step, thres, stream())
        torch.cuda.synchronize()
        s_r = orc.fold4(np.where(masks[step], sc.numpy(), np.float16(0)))
        widx, wkeep = orc.build_index_dec(s_r, thres)
        assert np.array_equal(idx.cpu().numpy(), widx)
        assert np.array_equal(wkeep, keeps[step])
        k = int(totals_d.cpu().numpy()[step])
        assert k == len(want_syms[step])
        assert np.array_equal(cidx.cpu().numpy()[base:base + k], (want_syms[step] & 0xff).astype(np.uint8))
        dec = (want_syms[step] >> 8).astype(np.int8)
        decoded[base:base + k] = torch.from_numpy(dec).to(dev)
        md = mean_list[step].to(dev)
        call(ops.y_step_dec_restore, ptr(decoded), ptr(cond), ptr(cnt), ptr(totals_d), ptr(md), C,
             ptr(acc_d), C, H, W, C, step, stream())
        torch.cuda.synchronize()
        base += k
    assert np.array_equal(acc_d.cpu().numpy(), want_acc), "decoder y_hat_so_far == encoder's"


@pytest.mark.parametrize("nsteps,H,W,C,thres", [(2, 17, 30, 128, 0.15), (4, 8, 8, 256, 0.0), (2, 68, 120, 128, 0.15),
                                                (4, 68, 120, 256, 0.15), (4, 5, 3, 256, 0.15)])
def test_mask_steps_exact(ops, nsteps, H, W, C, thres):
    """The inter models' full-tensor masked steps (LD: 2, HT-S: 4): divide, quantise step by step
    against changing means, final multiply, ONE index/compaction over all channels; then the
    decoder side from the "decoded" symbols - bit-exact against oracle/symbols_np.py, operands as
    channel slices of a wider buffer (the codecs' concatenation views)."""
    from gpu_util import call, ptr, stream
    from oracle import symbols_np as orc
    dev = "cuda"
    y = _rand((H, W, C), 6.0, 71)
    common = torch.zeros((H, W, 3 * C), dtype=torch.half)
    common[..., :C] = (_rand((H, W, C), 0.6, 72) + 0.9).half()              # q_dec, some below 0.5
    common[0, 0, :4] = torch.tensor([0.25, 0.5, -1.0, 3.0]).half()
    common[..., C:2 * C] = (_rand((H, W, C), 1.0, 73).float().exp() * 0.3).half()   # scales
    common[..., 2 * C:] = _rand((H, W, C), 2.0, 74)                          # means of step 0
    later_means = [_rand((H, W, C), 2.0, 75 + i) for i in range(nsteps - 1)]
    q_np, sc_np = common[..., :C].numpy(), common[..., C:2 * C].numpy()
    masks = orc.get_mask_2x(H, W, C) if nsteps == 2 else orc.get_mask_4x(H, W, C)
    # oracle
    y_div = orc.divide_with_clamp(y.numpy(), q_np)
    y_q_all = np.zeros((H, W, C), np.float16)
    y_hat = np.zeros((H, W, C), np.float16)
    for k in range(nsteps):
        mk = common[..., 2 * C:].numpy() if k == 0 else later_means[k - 1].numpy()
        yq, yh = orc.process_with_mask_2x(y_div, sc_np, mk, masks[k], thres)
        y_q_all = (y_q_all + yq).astype(np.float16)
        y_hat = (y_hat + yh).astype(np.float16)
    y_hat = (y_hat * orc.clamp_min_half(q_np)).astype(np.float16)
    comb, keep = orc.build_index_enc(y_q_all, sc_np, thres)
    # device, encoder side
    n = H * W * C
    yd, cd = y.to(dev), common.to(dev)
    cat = torch.full((H, W, 2 * C), 5.0, dtype=torch.half, device=dev)     # y_hat lives in channels 0..C-1
    sym = torch.zeros(n, dtype=torch.int16, device=dev)
    cond = torch.zeros((n + 7) // 8 + 8, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(ops.symbol_blocks(n), dtype=torch.int32, device=dev)
    comp = torch.zeros(n, dtype=torch.int16, device=dev)
    totals = torch.zeros(4, dtype=torch.int32, device=dev)
    eb = 2      # bytes per fp16
    for k in range(nsteps):
        if k == 0:
            mp, ldm = cd.data_ptr() + 2 * C * eb, 3 * C
        else:
            md = later_means[k - 1].to(dev)
            mp, ldm = md.data_ptr(), C
        call(ops.mask_step_enc, ptr(yd), C, ptr(cd), 3 * C, ctypes.c_void_p(cd.data_ptr() + C * eb), 3 * C,
             ctypes.c_void_p(mp), ldm, ptr(cat), 2 * C, ptr(sym), ptr(cond), ptr(cnt), ptr(comp), ptr(totals),
             H, W, C, nsteps, k, thres, stream())
        torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), y_div), "y / max(q, 0.5)"
    assert np.array_equal(sym.cpu().numpy(), comb)
    assert np.array_equal(cat[..., :C].cpu().numpy(), y_hat)
    assert (cat[..., C:] == 5.0).all(), "neighbouring channels untouched"
    k_enc = int(totals.cpu()[0])
    assert k_enc == int(keep.sum()) and np.array_equal(comp.cpu().numpy()[:k_enc], comb[keep])
    # decoder side
    idx = torch.zeros(n, dtype=torch.uint8, device=dev)
    cidx = torch.zeros(n, dtype=torch.uint8, device=dev)
    totals_d = torch.zeros(4, dtype=torch.int32, device=dev)
    call(ops.mask_dec_index, ctypes.c_void_p(cd.data_ptr() + C * eb), 3 * C, ptr(idx), ptr(cond), ptr(cnt),
         ptr(cidx), ptr(totals_d), H, W, C, thres, stream())
    torch.cuda.synchronize()
    widx, wkeep = orc.build_index_dec(sc_np, thres)
    assert np.array_equal(idx.cpu().numpy(), widx) and int(totals_d.cpu()[0]) == k_enc
    assert np.array_equal(cidx.cpu().numpy()[:k_enc], widx[wkeep])
    decoded = torch.from_numpy((comb[keep] >> 8).astype(np.int8)).to(dev) if k_enc else torch.zeros(1, dtype=torch.int8, device=dev)
    yqs = torch.zeros(n, dtype=torch.int8, device=dev)
    cat_d = torch.full((H, W, 2 * C), -3.0, dtype=torch.half, device=dev)
    for k in range(nsteps):
        if k == 0:
            mp, ldm = cd.data_ptr() + 2 * C * eb, 3 * C
        else:
            md = later_means[k - 1].to(dev)
            mp, ldm = md.data_ptr(), C
        call(ops.mask_step_dec, ptr(decoded), ptr(cond), ptr(cnt), ptr(totals_d), ptr(yqs), ctypes.c_void_p(mp), ldm,
             ptr(cd), 3 * C, ptr(cat_d), 2 * C, H, W, C, nsteps, k, stream())
        torch.cuda.synchronize()
    assert np.array_equal(cat_d[..., :C].cpu().numpy(), y_hat), "decoder y_hat == encoder's"


@pytest.mark.parametrize("P,C,CF,res2,quant,q2,inplace", [
    (300, 384, 384, False, False, False, True),
    (1024, 256, 128, True, False, False, True),
    (130, 128, 64, False, True, True, False),
    (2040, 384, 384, True, False, True, False),
    (32640, 384, 384, False, False, True, True),
    (32640, 256, 128, False, True, False, True),
])
def test_ffn_fused_equals_two_launches(ops, P, C, CF, res2, quant, q2, inplace):
    """ffn.0 + ffn.2 in one launch == conv1x1(wsilu, chunk_add) followed by conv1x1(residuals, quant),
    bit for bit (the two-launch kernels are themselves checked against the oracle above)."""
    from gpu_util import call, ptr, stream
    dev = "cuda"
    ld = C + 64                                            # a channel slice of a wider buffer
    xbuf = _rand((P, ld), 1.0, 91).to(dev)
    w0 = (_rand((4 * CF, C), 1.0, 92) / C ** 0.5).half().to(dev)
    b0 = _rand((4 * CF,), 0.3, 93).to(dev)
    w2 = (_rand((C, CF), 1.0, 94) / CF ** 0.5).half().to(dev)
    b2 = _rand((C,), 0.3, 95).to(dev)
    r2 = _rand((P, C), 1.0, 96).to(dev) if res2 else None
    q = (_rand((C,), 0.2, 97) + 1.0).half().to(dev) if quant else None
    qq = (_rand((C,), 0.2, 98) + 1.0).half().to(dev) if q2 else None
    # two launches
    t = torch.zeros((P, CF), dtype=torch.half, device=dev)
    want = torch.zeros((P, C), dtype=torch.half, device=dev)
    call(ops.conv1x1, ptr(xbuf), ld, ptr(w0), ptr(b0), None, 0, None, 0, None, None, ptr(t), CF, P, C, 4 * CF, 3, stream())
    call(ops.conv1x1, ptr(t), CF, ptr(w2), ptr(b2), ptr(xbuf), ld, ptr(r2), C, ptr(q), ptr(qq), ptr(want), C, P, CF, C, 0, stream())
    torch.cuda.synchronize()
    # one launch
    if inplace:
        ybuf, ldy = xbuf.clone(), ld
        xin = ybuf
    else:
        ybuf, ldy = torch.full((P, C + 8), 9.0, dtype=torch.half, device=dev), C + 8
        xin = xbuf
    call(ops.ffn_fused, ptr(xin), ld, ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(r2), C, ptr(q), ptr(qq), ptr(ybuf), ldy,
         P, C, CF, stream())
    torch.cuda.synchronize()
    got = ybuf[:, :C]
    bad = int((got != want).sum())
    assert bad == 0, "%d of %d outputs differ" % (bad, want.numel())
    if inplace:
        assert torch.equal(ybuf[:, C:], xbuf[:, C:]), "channels beyond C untouched"
    else:
        assert (ybuf[:, C:] == 9.0).all()


@pytest.mark.parametrize("H,W,C,CD,CF,use_dw,shortcut,quant,q2,dc0", [
    (8, 16, 256, 128, 128, True, False, False, False, False),
    (13, 37, 256, 128, 128, True, True, False, False, False),
    (24, 40, 128, 64, 64, True, False, True, True, False),
    (17, 30, 256, 64, 192, False, False, True, False, False),
    (136, 240, 256, 128, 128, True, False, False, True, False),
    (136, 240, 128, 64, 64, True, True, False, False, False),
    (8, 16, 256, 128, 128, True, False, False, False, True),
    (13, 37, 256, 128, 128, True, True, False, False, True),
    (24, 40, 128, 64, 64, True, False, True, True, True),
    (136, 240, 256, 128, 128, True, False, False, True, True),
    (136, 240, 128, 64, 64, True, True, False, False, True),
    # round 6: FULL-width 128-wide blocks (the intra model's hyper networks at / 16 .. / 64: five launches each until then)
    (68, 120, 128, 128, 128, True, False, False, False, True),
    (34, 60, 128, 128, 128, True, True, False, False, True),
    (17, 30, 128, 128, 128, True, True, True, True, False),
    (13, 37, 128, 128, 128, False, False, False, False, False),
    (4, 4, 128, 128, 128, True, True, False, False, True),
])
def test_dcb_tail_equals_four_launches(ops, H, W, C, CD, CF, use_dw, shortcut, quant, q2, dc0):
    """depthwise + dc.3 + ffn.0 + ffn.2 in one launch == dwconv3x3, conv1x1(residual),
    conv1x1(wsilu, chunk_add), conv1x1(residuals, quant) one after the other, bit for bit, on
    pictures that do and do not divide into 8x16 patches."""
    from gpu_util import call, ptr, stream
    dev = "cuda"
    P = H * W
    t1 = _rand((P, CD), 1.0, 101).to(dev)
    dww = _rand((9, CD), 0.4, 102).to(dev)
    ldx = C + 32
    xbuf = _rand((P, ldx), 1.0, 103).to(dev)
    w3 = (_rand((C, CD), 1.0, 104) / CD ** 0.5).half().to(dev)
    b3 = _rand((C,), 0.3, 105).to(dev)
    w0 = (_rand((4 * CF, C), 1.0, 106) / C ** 0.5).half().to(dev)
    b0 = _rand((4 * CF,), 0.3, 107).to(dev)
    w2 = (_rand((C, CF), 1.0, 108) / CF ** 0.5).half().to(dev)
    b2 = _rand((C,), 0.3, 109).to(dev)
    q = (_rand((C,), 0.2, 110) + 1.0).half().to(dev) if quant else None
    qq = (_rand((C,), 0.2, 111) + 1.0).half().to(dev) if q2 else None
    if shortcut and quant:
        pytest.skip("quant with two residuals is not a reference op")
    w1 = (_rand((CD, C), 1.0, 112) / C ** 0.5).half().to(dev)
    b1 = _rand((CD,), 0.3, 113).to(dev)
    if dc0:         # t1 = dc.0 of the block input (five launches in total)
        call(ops.conv1x1, ptr(xbuf), ldx, ptr(w1), ptr(b1), None, 0, None, 0, None, None, ptr(t1), CD, P, C, CD, 1, stream())
    # four launches
    t2 = torch.zeros((P, CD), dtype=torch.half, device=dev)
    if use_dw:
        call(ops.dwconv3x3, ptr(t1), CD, ptr(dww), ptr(t2), CD, H, W, CD, stream())
    else:
        t2.copy_(t1)
    y1 = torch.zeros((P, C), dtype=torch.half, device=dev)
    call(ops.conv1x1, ptr(t2), CD, ptr(w3), ptr(b3), ptr(xbuf), ldx, None, 0, None, None, ptr(y1), C, P, CD, C, 0, stream())
    t3 = torch.zeros((P, CF), dtype=torch.half, device=dev)
    call(ops.conv1x1, ptr(y1), C, ptr(w0), ptr(b0), None, 0, None, 0, None, None, ptr(t3), CF, P, C, 4 * CF, 3, stream())
    want = torch.zeros((P, C), dtype=torch.half, device=dev)
    call(ops.conv1x1, ptr(t3), CF, ptr(w2), ptr(b2), ptr(y1), C, ptr(xbuf) if shortcut else None, ldx, ptr(q), ptr(qq),
         ptr(want), C, P, CF, C, 0, stream())
    torch.cuda.synchronize()
    # one launch: in place on the block buffer (as the codec runs the tail), out of place with dc.0 inside
    ybuf = xbuf.clone()
    xin = xbuf if dc0 else ybuf
    call(ops.dcb_tail, ptr(w1) if dc0 else None, ptr(b1) if dc0 else None, None if dc0 else ptr(t1), CD,
         ptr(dww) if use_dw else None, ptr(xin), ldx, ptr(w3), ptr(b3), ptr(w0), ptr(b0),
         ptr(w2), ptr(b2), ptr(q), ptr(qq), ptr(ybuf), ldx, H, W, C, CD, CF, 1 if shortcut else 0, stream())
    torch.cuda.synchronize()
    bad = int((ybuf[:, :C] != want).sum())
    assert bad == 0, "%d of %d outputs differ" % (bad, want.numel())
    assert torch.equal(ybuf[:, C:], xbuf[:, C:]), "channels beyond C untouched"


@pytest.mark.parametrize("C,CI,P,shortcut,quant,q2,nxt,inplace", [
    (384, 384, 64, False, False, False, False, False),   # one 64-pixel workgroup
    (384, 384, 100, False, False, False, True, True),    # 32-pixel workgroups, ragged last one, in place, next dc.0
    (384, 384, 13000, True, False, True, False, False),  # 64-pixel workgroups, ragged; block shortcut + scale after rounding
    (384, 384, 32640, False, True, False, True, True),   # 1080p P8 grid (510 tiles), fused quant, the chain configuration
    (512, 512, 8160, False, False, False, True, False),  # 1080p P16 grid of the prior networks (255 workgroups of 32 pixels)
    (512, 512, 777, True, False, True, True, False),     # shortcut + scale after rounding + next dc.0, ragged
    (512, 512, 2000, False, True, False, False, True),   # fused quant, in place
    (512, 512, 32640, False, False, False, False, True), # 512-channel full-width blocks at P8
    (512, 256, 32640, False, False, False, True, True),  # the hierarchical models' dcb2 chains at P8 (layers.py:128-159)
    (512, 256, 777, True, False, True, True, False),     # ... 32-pixel workgroups, ragged, both residuals
    (512, 256, 13000, False, True, False, False, False), # ... fused quant, ragged 64-pixel workgroups
    (256, 128, 32640, False, False, False, True, True),  # the low-delay model's dcb2 chains at P8
    (256, 128, 100, True, False, True, False, False),    # ... 32-pixel workgroups (the WSiLU table behind LDS padding)
    (256, 128, 20000, False, True, False, True, False),  # ... ragged, fused quant, next dc.0
    (256, 256, 32640, False, False, False, True, True),  # the hierarchical models' reconstruction heads at P8
    (256, 256, 510, True, False, True, False, False),    # ... their hyper networks at / 64
    (768, 768, 8160, False, False, False, True, True),   # the hierarchical models' prior fusion at / 16 (32-pixel workgroups only)
    (768, 768, 32400, True, False, True, False, False),  # ... at 3840x2160: still 32 pixels per workgroup (LDS)
    # just below the 64-pixel threshold (1024x768 and the like: 12 288 pixels at / 8): 32-pixel tiles, MORE tiles than CUs,
    # so the persistent workgroups walk several 32-pixel tiles each
    (384, 384, 12288, False, True, False, True, True),
    (512, 256, 12000, True, False, True, True, False),
    (512, 512, 12700, False, False, False, True, True),
    (256, 256, 12288, False, True, False, True, False),
    (256, 128, 12001, True, False, False, True, False),
    # round 4: the 256-wide blocks on large ragged grids and on the 3840x2160 grid (129 600 pixels: 8 tiles per persistent
    # workgroup). (Written for a 128-pixel-per-workgroup form of the 8-wave kernel that was bit-identical and no faster -
    # 27.8 vs 27.6 us at (256, 128), 44.8 vs 44.1 at (256, 256) - and was dropped: DESIGN.md 5, round 4.)
    (256, 128, 30001, True, False, True, True, False),
    (256, 128, 129600, False, True, False, True, True),
    (256, 256, 30001, False, True, False, True, False),
    (256, 256, 27000, True, False, True, False, False),
    (256, 256, 129600, False, False, False, True, True),
    # round 6: the low-delay model's prior fusion blocks: inner width 192 (LDS rows padded to 256 channels, ffn.0's tile pairs
    # 2 | 1 over the waves of a SIMD, dc.0's six tiles on six of the eight waves)
    (384, 192, 8160, False, False, False, True, True),
    (384, 192, 8160, True, False, True, False, False),
    (384, 192, 32640, False, True, False, True, False),
    (384, 192, 12289, False, False, True, True, True),
    (384, 192, 33, True, False, False, True, False),
    # ... and the intra decoder's last block: block width 192 as well (six tiles of the C-wide layers on six waves, two waves on zeros)
    (192, 192, 32640, False, False, False, False, False),
    (192, 192, 32640, True, False, True, True, False),
    (192, 192, 12289, False, True, False, True, True),
    (192, 192, 100, False, False, False, False, True),
    (192, 192, 129600, False, False, False, False, False),
])
def test_dcb_nsplit_equals_launch_sequence(ops, C, CI, P, shortcut, quant, q2, nxt, inplace):
    """The same block through kernels/dcb_nsplit.hip (round 3: activations in LDS, every wave streams its quarter of
    every weight matrix from a packed copy) == the launch sequence conv1x1(dc.3, residual) -> conv1x1(ffn.0, wsilu,
    chunk_add) -> conv1x1(ffn.2, residuals, quant) (-> conv1x1(dc.0, wsilu)), bit for bit, for full- and half-width
    blocks (CI = inner width), both workgroup sizes, ragged grids, in place and out of place."""
    from gpu_util import call, ptr, stream
    dev = "cuda"
    ldx = C + 64
    xbuf = _rand((P, ldx), 1.0, 401).to(dev)
    t2 = _rand((P, CI), 1.0, 402).to(dev)
    w3 = (_rand((C, CI), 1.0, 403) / CI ** 0.5).half().to(dev)
    b3 = _rand((C,), 0.3, 404).to(dev)
    w0 = (_rand((4 * CI, C), 1.0, 405) / C ** 0.5).half().to(dev)
    b0 = _rand((4 * CI,), 0.3, 406).to(dev)
    w2 = (_rand((C, CI), 1.0, 407) / CI ** 0.5).half().to(dev)
    b2 = _rand((C,), 0.3, 408).to(dev)
    w1 = (_rand((CI, C), 1.0, 409) / C ** 0.5).half().to(dev)
    b1 = _rand((CI,), 0.3, 410).to(dev)
    q = (_rand((C,), 0.2, 411) + 1.0).half().to(dev) if quant else None
    qq = (_rand((C,), 0.2, 412) + 1.0).half().to(dev) if q2 else None
    y1 = torch.zeros((P, C), dtype=torch.half, device=dev)
    t = torch.zeros((P, CI), dtype=torch.half, device=dev)
    want = torch.zeros((P, C), dtype=torch.half, device=dev)
    want_t1 = torch.zeros((P, CI), dtype=torch.half, device=dev)
    call(ops.conv1x1, ptr(t2), CI, ptr(w3), ptr(b3), ptr(xbuf), ldx, None, 0, None, None, ptr(y1), C, P, CI, C, 0, stream())
    call(ops.conv1x1, ptr(y1), C, ptr(w0), ptr(b0), None, 0, None, 0, None, None, ptr(t), CI, P, C, 4 * CI, 3, stream())
    call(ops.conv1x1, ptr(t), CI, ptr(w2), ptr(b2), ptr(y1), C, ptr(xbuf) if shortcut else None, ldx, ptr(q), ptr(qq),
         ptr(want), C, P, CI, C, 0, stream())
    call(ops.conv1x1, ptr(want), C, ptr(w1), ptr(b1), None, 0, None, 0, None, None, ptr(want_t1), CI, P, C, CI, 1, stream())
    torch.cuda.synchronize()

    def run(op, *inner):
        if inplace:
            ybuf, ldy = xbuf.clone(), ldx
            xin = ybuf
        else:
            ybuf, ldy = torch.full((P, C + 8), 9.0, dtype=torch.half, device=dev), C + 8
            xin = xbuf
        t1 = torch.full((P, CI + 8), 7.0, dtype=torch.half, device=dev)
        call(op, ptr(t2), CI, ptr(xin), ldx, ptr(w3), ptr(b3), ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(q), ptr(qq),
             ptr(w1) if nxt else None, ptr(b1) if nxt else None, ptr(t1) if nxt else None, CI + 8, ptr(ybuf), ldy,
             P, C, *inner, 1 if shortcut else 0, stream())
        torch.cuda.synchronize()
        return ybuf, t1

    ybuf, t1 = run(ops.dcb_nsplit, CI)
    bad = int((ybuf[:, :C] != want).sum())
    assert bad == 0, "y: %d of %d outputs differ" % (bad, want.numel())
    if inplace:
        assert torch.equal(ybuf[:, C:], xbuf[:, C:]), "channels beyond C untouched"
    else:
        assert (ybuf[:, C:] == 9.0).all()
    if nxt:
        bad = int((t1[:, :CI] != want_t1).sum())
        assert bad == 0, "next dc.0: %d of %d outputs differ" % (bad, want_t1.numel())
        assert (t1[:, CI:] == 7.0).all()
    else:
        assert (t1 == 7.0).all()


@pytest.mark.parametrize("C,CI,NN,P,shortcut,quant,qfin,inplace", [
    (512, 512, 512, 8160, False, False, False, True),    # intra / HT-L: y_prior_fusion.conv.3, y_spatial_prior.conv.3 at / 16
    (512, 512, 512, 32640, False, True, False, False),   # ... at 3840x2160 (64-pixel workgroups)
    (512, 512, 256, 8160, False, False, False, True),    # HT-S: y_spatial_prior.conv.3 (256 of 512 channels out)
    (512, 512, 256, 13001, True, False, True, False),    # ... ragged, 64-pixel workgroups, with a scale
    (768, 768, 768, 8160, False, False, False, True),    # hierarchical models: y_prior_fusion.conv.3
    (768, 768, 768, 777, True, False, True, False),
    (256, 128, 128, 8160, False, False, False, True),    # LD: y_spatial_prior.conv.2 at / 16 (the upper waves have no tile)
    (256, 128, 128, 32641, False, True, False, False),
    (256, 128, 256, 32640, False, False, True, True),    # LD: decoder.conv2 with its quant scale at / 8
    (256, 128, 256, 100, True, False, True, False),
    (256, 128, 192, 32640, False, False, False, True),   # LD: recon_head.head (6 tiles: two of the upper waves idle)
    (256, 128, 192, 12001, True, False, False, False),
    (256, 128, 192, 20001, False, True, False, False),
    (256, 128, 192, 70, False, False, True, False),
    (256, 256, 192, 32640, False, False, False, True),   # hierarchical models: the 8 reconstruction heads
    (256, 256, 192, 129600, False, False, False, True),  # ... at 3840x2160
    (256, 256, 192, 513, True, False, True, False),
    (256, 256, 192, 13513, False, True, True, False),
    (384, 192, 384, 8160, False, False, False, True),    # LD: y_prior_fusion.conv.3 behind the (384, 192) blocks at / 16
    (384, 192, 384, 32640, False, True, False, False),   # ... at 3840x2160
    (384, 192, 384, 45, True, False, True, False),
])
def test_dcb_nsplit_closing_conv_equals_launch_sequence(ops, C, CI, NN, P, shortcut, quant, qfin, inplace):
    """Round 6: the 1x1 conv that CLOSES a chain of blocks inside the last block's launch (the NEXT slot of the 8-wave kernel)
    == the block's launch sequence followed by conv1x1(bias [, quant]) on its output, bit for bit; the block's own output
    is still written."""
    from gpu_util import call, ptr, stream
    dev = "cuda"
    if not ops.dcb_nsplit_fin_supported(C, CI, NN):
        pytest.skip("no kernel variant (4-wave kernel selected)")
    ldx = C + 64
    xbuf = _rand((P, ldx), 1.0, 601).to(dev)
    t2 = _rand((P, CI), 1.0, 602).to(dev)
    w3 = (_rand((C, CI), 1.0, 603) / CI ** 0.5).half().to(dev)
    b3 = _rand((C,), 0.3, 604).to(dev)
    w0 = (_rand((4 * CI, C), 1.0, 605) / C ** 0.5).half().to(dev)
    b0 = _rand((4 * CI,), 0.3, 606).to(dev)
    w2 = (_rand((C, CI), 1.0, 607) / CI ** 0.5).half().to(dev)
    b2 = _rand((C,), 0.3, 608).to(dev)
    wf = (_rand((NN, C), 1.0, 609) / C ** 0.5).half().to(dev)
    bf = _rand((NN,), 0.3, 610).to(dev)
    q = (_rand((C,), 0.2, 611) + 1.0).half().to(dev) if quant else None
    qf = (_rand((NN,), 0.2, 612) + 1.0).half().to(dev) if qfin else None
    y1 = torch.zeros((P, C), dtype=torch.half, device=dev)
    t = torch.zeros((P, CI), dtype=torch.half, device=dev)
    want = torch.zeros((P, C), dtype=torch.half, device=dev)
    want_f = torch.zeros((P, NN), dtype=torch.half, device=dev)
    call(ops.conv1x1, ptr(t2), CI, ptr(w3), ptr(b3), ptr(xbuf), ldx, None, 0, None, None, ptr(y1), C, P, CI, C, 0, stream())
    call(ops.conv1x1, ptr(y1), C, ptr(w0), ptr(b0), None, 0, None, 0, None, None, ptr(t), CI, P, C, 4 * CI, 3, stream())
    call(ops.conv1x1, ptr(t), CI, ptr(w2), ptr(b2), ptr(y1), C, ptr(xbuf) if shortcut else None, ldx, ptr(q), None,
         ptr(want), C, P, CI, C, 0, stream())
    call(ops.conv1x1, ptr(want), C, ptr(wf), ptr(bf), None, 0, None, 0, ptr(qf), None, ptr(want_f), NN, P, C, NN, 0, stream())
    torch.cuda.synchronize()
    if inplace and not shortcut:
        ybuf, ldy = xbuf.clone(), ldx
        xin = ybuf
    else:
        ybuf, ldy = torch.full((P, C + 8), 9.0, dtype=torch.half, device=dev), C + 8
        xin = xbuf
    yf = torch.full((P, NN + 8), 7.0, dtype=torch.half, device=dev)
    call(ops.dcb_nsplit_fin, ptr(t2), CI, ptr(xin), ldx, ptr(w3), ptr(b3), ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(q), None,
         ptr(wf), ptr(bf), ptr(qf), ptr(yf), NN + 8, NN, ptr(ybuf), ldy, P, C, CI, 1 if shortcut else 0, stream())
    torch.cuda.synchronize()
    bad = int((ybuf[:, :C] != want).sum())
    assert bad == 0, "y: %d of %d outputs differ" % (bad, want.numel())
    bad = int((yf[:, :NN] != want_f).sum())
    assert bad == 0, "closing conv: %d of %d outputs differ" % (bad, want_f.numel())
    assert (yf[:, NN:] == 7.0).all(), "channels beyond the closing conv's width untouched"
    if not shortcut:
        # the codecs' form: nothing else reads the block's own output, so it is not stored at all (y = NULL)
        yf2 = torch.full((P, NN + 8), 7.0, dtype=torch.half, device=dev)
        call(ops.dcb_nsplit_fin, ptr(t2), CI, ptr(xbuf), ldx, ptr(w3), ptr(b3), ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(q), None,
             ptr(wf), ptr(bf), ptr(qf), ptr(yf2), NN + 8, NN, None, 0, P, C, CI, 0, stream())
        torch.cuda.synchronize()
        assert torch.equal(yf2, yf), "closing conv without the block's own output"


DW_CASES = [
    (135, 240, True, 0, False, False, False),     # LD at 1920x1080, / 8: 64-pixel workgroups, dc.0 of the next block behind ffn.2
    (136, 240, True, 0, True, True, False),
    (135, 241, False, 0, False, False, False),    # ragged last tile, rows that end inside a tile, nothing behind ffn.2
    (68, 120, True, 0, False, False, False),      # / 16: 32-pixel workgroups
    (67, 121, False, 0, True, False, False),
    (68, 120, False, 128, False, False, False),   # y_spatial_prior.conv.2 behind the block
    (135, 240, False, 256, False, False, True),   # decoder.conv2 with its quant scale
    (135, 240, False, 192, False, True, False),   # recon_head.head
    (270, 480, True, 0, False, False, False),     # 3840x2160, / 8: several tiles per workgroup
    (201, 65, True, 0, False, False, False),      # a picture barely wider than a tile
    (300, 50, True, 0, False, True, False),       # ... narrower: a tile spans rows
    (1000, 13, False, 192, False, False, False),  # ... five rows and more
    (9, 7, True, 0, False, False, False),         # one ragged 32-pixel... two tiles
    (1, 40, False, 0, False, False, False),       # a single row: no tap above or below
    (40, 1, True, 0, False, False, False),        # a single column: no tap left or right
    (3, 3, False, 128, False, False, True),
]
DW_CASES = [(256, 128) + c for c in DW_CASES] + [
    (384, 192, 68, 120, True, 0, False, False, False),     # LD's prior fusion at 1920x1080, / 16 (rows of 24 chunks: 2 2/3 pixels per wave and step)
    (384, 192, 68, 120, False, 384, False, False, False),  # ... its last block with y_prior_fusion.conv.3 behind it
    (384, 192, 67, 121, False, 0, True, True, False),
    (384, 192, 45, 80, True, 0, False, False, False),      # 1280x720
    (384, 192, 400, 31, True, 0, False, True, False),      # a tile spans rows
    (384, 192, 1, 33, False, 384, False, False, True),
    (384, 192, 70, 1, True, 0, False, False, False),
]


@pytest.mark.parametrize("C,CI,H,W,nxt,NN,shortcut,quant,qfin", DW_CASES)
def test_dcb_nsplit_with_depthwise_inside_equals_launch_sequence(ops, C, CI, H, W, nxt, NN, shortcut, quant, qfin):
    """Round 6: the (256, 128) / (384, 192) block launch with the block's depthwise 3x3 conv inside (dc.0's output around a tile by LDS-DMA,
    the conv by the waves that idle in the NEXT slot) == dwconv3x3 followed by the block launch on its output, bit for bit -
    with dc.0 of the next block, a closing conv or nothing behind ffn.2."""
    from gpu_util import call, ptr, stream
    dev = "cuda"
    P = H * W
    if not ops.dcb_nsplit_dw_supported(C, CI, P):
        pytest.skip("DCVC_NSPLIT_DW=0")
    ldx = C + 64
    xbuf = _rand((P, ldx), 1.0, 901).to(dev)
    t1 = _rand((P, CI), 1.0, 902).to(dev)
    wd = _rand((CI, 1, 3, 3), 0.3, 913).to(dev)
    wt = wd[:, 0].permute(1, 2, 0).reshape(9, CI).contiguous()
    w3 = (_rand((C, CI), 1.0, 903) / CI ** 0.5).half().to(dev)
    b3 = _rand((C,), 0.3, 904).to(dev)
    w0 = (_rand((4 * CI, C), 1.0, 905) / C ** 0.5).half().to(dev)
    b0 = _rand((4 * CI,), 0.3, 906).to(dev)
    w2 = (_rand((C, CI), 1.0, 907) / CI ** 0.5).half().to(dev)
    b2 = _rand((C,), 0.3, 908).to(dev)
    w1n = (_rand((CI, C), 1.0, 909) / C ** 0.5).half().to(dev) if nxt else None
    b1n = _rand((CI,), 0.3, 910).to(dev) if nxt else None
    wf = (_rand((NN, C), 1.0, 914) / C ** 0.5).half().to(dev) if NN else None
    bf = _rand((NN,), 0.3, 915).to(dev) if NN else None
    q = (_rand((C,), 0.2, 911) + 1.0).half().to(dev) if quant else None
    qf = (_rand((NN,), 0.2, 912) + 1.0).half().to(dev) if qfin else None
    # the launches the codecs ran until now: depthwise conv, then the block launch on its output
    t2 = torch.zeros((P, CI), dtype=torch.half, device=dev)
    call(ops.dwconv3x3, ptr(t1), CI, ptr(wt), ptr(t2), CI, H, W, CI, stream())
    want = torch.full((P, C + 8), 9.0, dtype=torch.half, device=dev)
    want_n = torch.full((P, CI + 8), 7.0, dtype=torch.half, device=dev)
    want_f = torch.full((P, NN + 8), 7.0, dtype=torch.half, device=dev)
    if NN:
        call(ops.dcb_nsplit_fin, ptr(t2), CI, ptr(xbuf), ldx, ptr(w3), ptr(b3), ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(q), None,
             ptr(wf), ptr(bf), ptr(qf), ptr(want_f), NN + 8, NN, ptr(want), C + 8, P, C, CI, 1 if shortcut else 0, stream())
    else:
        call(ops.dcb_nsplit, ptr(t2), CI, ptr(xbuf), ldx, ptr(w3), ptr(b3), ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(q), None,
             ptr(w1n), ptr(b1n), ptr(want_n) if nxt else None, CI + 8, ptr(want), C + 8, P, C, CI, 1 if shortcut else 0, stream())
    torch.cuda.synchronize()
    from oracle import nn
    if P <= 40000:
        assert np.array_equal(t2.cpu().numpy().reshape(H, W, CI), nn.dwconv3x3(t1.cpu().numpy().reshape(H, W, CI), wd.cpu().numpy()))
    y = torch.full((P, C + 8), 9.0, dtype=torch.half, device=dev)
    t1n = torch.full((P, CI + 8), 7.0, dtype=torch.half, device=dev)
    yf = torch.full((P, NN + 8), 7.0, dtype=torch.half, device=dev)
    for rep in range(2):        # (twice: the second launch finds the first one's leftovers in LDS-sized caches, not in its logic)
        call(ops.dcb_nsplit_dw, ptr(t1), CI, ptr(wt), W, ptr(xbuf), ldx, ptr(w3), ptr(b3), ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(q), None,
             ptr(w1n), ptr(b1n), ptr(t1n) if nxt else None, CI + 8,
             ptr(wf), ptr(bf), ptr(qf), ptr(yf) if NN else None, NN + 8, NN,
             ptr(y), C + 8, P, C, CI, 1 if shortcut else 0, stream())
        torch.cuda.synchronize()
        bad = int((y != want).sum())
        assert bad == 0, "y: %d of %d outputs differ" % (bad, want.numel())
        if nxt:
            bad = int((t1n != want_n).sum())
            assert bad == 0, "next block's dc.0: %d of %d outputs differ" % (bad, want_n.numel())
        if NN:
            bad = int((yf != want_f).sum())
            assert bad == 0, "closing conv: %d of %d outputs differ" % (bad, want_f.numel())


def test_dcb_nsplit_with_depthwise_inside_refuses_what_it_cannot_run(ops):
    from gpu_util import ptr, stream
    dev = "cuda"
    if not ops.dcb_nsplit_dw_supported(256, 128, 64):
        pytest.skip("DCVC_NSPLIT_DW=0")
    assert ops.dcb_nsplit_dw_supported(384, 384, 8160) == 0
    assert ops.dcb_nsplit_dw_supported(384, 192, 8160) == 1 and ops.dcb_nsplit_dw_supported(384, 192, 32640) == 0      # 64-pixel workgroups: no room
    z = torch.zeros((512, 1024), dtype=torch.half, device=dev)      # (large enough for ffn.0's weights: the entry point packs them before the launch is checked)
    # dc.0's output for the next block into the buffer the launch still reads its own from
    rc = ops.dcb_nsplit_dw(ptr(z), 128, ptr(z), 8, ptr(z), 256, ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), None, None,
                           ptr(z), ptr(z), ptr(z), 128, None, None, None, None, 0, 0, ptr(z), 256, 64, 256, 128, 0, stream())
    assert rc != 0
    # a width that does not divide the pixel count
    rc = ops.dcb_nsplit_dw(ptr(z), 128, ptr(z), 7, ptr(z), 256, ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), None, None,
                           None, None, None, 0, None, None, None, None, 0, 0, ptr(z), 256, 64, 256, 128, 0, stream())
    assert rc != 0
    # a block shape without such a variant
    rc = ops.dcb_nsplit_dw(ptr(z), 256, ptr(z), 8, ptr(z), 256, ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), None, None,
                           None, None, None, 0, None, None, None, None, 0, 0, ptr(z), 256, 64, 256, 256, 0, stream())
    assert rc != 0


@pytest.mark.parametrize("CIN,C,CI,P", [
    (448, 256, 128, 32640),    # LD: encoder.conv1.0 ([x unshuffled | ctx] -> 256) at / 8
    (448, 256, 128, 77),
    (512, 256, 128, 32640),    # LD: decoder.conv1.0, feature_adaptor_m.conv.0
    (512, 256, 128, 8160),     # LD: y_spatial_prior.conv.0 at / 16 (32-pixel workgroups)
    (512, 256, 128, 12801),    # ... ragged, 64-pixel workgroups
    (192, 256, 128, 32640),    # LD: feature_adaptor_i.conv.0 (input rows padded to 256 channels in LDS)
    (192, 256, 128, 100),
    (128, 256, 256, 8160),     # intra: hyper_dec.conv.2
    (512, 256, 256, 32640),    # hierarchical models: recon_head conv2 / conv .0
    (512, 256, 256, 513),
    (192, 384, 384, 32640),    # intra: enc.enc_1
    (192, 384, 384, 12289),
    (256, 512, 512, 8160),     # intra: y_prior_fusion.conv.0
    (512, 512, 512, 8160),     # intra / hierarchical: y_spatial_prior_adaptor_k
    (512, 512, 512, 32641),    # ... at 3840x2160
    (192, 512, 512, 32640),    # HT-L: feature_adaptor_i.conv.0
    (192, 512, 256, 32640),    # HT-S: feature_adaptor_i.conv.0
    (192, 512, 256, 45),
    (384, 192, 192, 32640),    # intra: dec.dec_2 (block width 192: two of the eight waves walk zeros)
    (384, 192, 192, 12801),
    (384, 192, 192, 70),
])
def test_dcb_pair_equals_two_launches(ops, CIN, C, CI, P):
    """Round 6: a block's adaptor and its dc.0 in ONE launch (kernels/dcb_pair8_kernel.h: the adaptor output stays in LDS as
    dc.0's operand) == conv1x1(bias) followed by conv1x1(bias, wsilu), bit for bit, with channel-slice views either side."""
    from gpu_util import call, ptr, stream
    dev = "cuda"
    if not ops.dcb_pair_supported(CIN, C, CI):
        pytest.skip("no kernel variant (4-wave kernel selected)")
    ldx = CIN + 64
    xbuf = _rand((P, ldx), 1.0, 701).to(dev)
    wa = (_rand((C, CIN), 1.0, 702) / CIN ** 0.5).half().to(dev)
    ba = _rand((C,), 0.3, 703).to(dev)
    w1 = (_rand((CI, C), 1.0, 704) / C ** 0.5).half().to(dev)
    b1 = _rand((CI,), 0.3, 705).to(dev)
    want_y = torch.zeros((P, C), dtype=torch.half, device=dev)
    want_t = torch.zeros((P, CI), dtype=torch.half, device=dev)
    call(ops.conv1x1, ptr(xbuf[:, 64:]), ldx, ptr(wa), ptr(ba), None, 0, None, 0, None, None, ptr(want_y), C, P, CIN, C, 0, stream())
    call(ops.conv1x1, ptr(want_y), C, ptr(w1), ptr(b1), None, 0, None, 0, None, None, ptr(want_t), CI, P, C, CI, 1, stream())
    y = torch.full((P, C + 8), 9.0, dtype=torch.half, device=dev)
    t1 = torch.full((P, CI + 8), 7.0, dtype=torch.half, device=dev)
    call(ops.dcb_pair, ptr(xbuf[:, 64:]), ldx, ptr(wa), ptr(ba), ptr(w1), ptr(b1), ptr(y), C + 8, ptr(t1), CI + 8, P, CIN, C, CI, stream())
    torch.cuda.synchronize()
    bad = int((y[:, :C] != want_y).sum())
    assert bad == 0, "adaptor: %d of %d outputs differ" % (bad, want_y.numel())
    bad = int((t1[:, :CI] != want_t).sum())
    assert bad == 0, "dc.0: %d of %d outputs differ" % (bad, want_t.numel())
    assert (y[:, C:] == 9.0).all() and (t1[:, CI:] == 7.0).all()


def test_dcb_nsplit_reads_the_weights_of_the_call(ops):
    """dcvc_dcb_nsplit packs the weights it is GIVEN, on every call: rewriting them in place between two calls (same
    pointers - round 3's pointer-keyed cache returned the first call's copy, advisor finding) changes the result
    accordingly; the handle form keeps the snapshot taken at pack time until it is packed again."""
    import ctypes
    from gpu_util import call, ptr, stream
    dev, C, CI, P = "cuda", 256, 128, 300
    x = _rand((P, C), 1.0, 501).to(dev)
    t2 = _rand((P, CI), 1.0, 502).to(dev)
    ws = [(_rand(shape, 1.0, 503 + i) / shape[1] ** 0.5).half().to(dev) for i, shape in enumerate(((C, CI), (4 * CI, C), (C, CI), (CI, C)))]
    bs = [_rand((n,), 0.3, 510 + i).to(dev) for i, n in enumerate((C, 4 * CI, C, CI))]

    def direct():
        y = torch.zeros((P, C), dtype=torch.half, device=dev)
        t1 = torch.zeros((P, CI), dtype=torch.half, device=dev)
        call(ops.dcb_nsplit, ptr(t2), CI, ptr(x), C, ptr(ws[0]), ptr(bs[0]), ptr(ws[1]), ptr(bs[1]), ptr(ws[2]), ptr(bs[2]), None, None,
             ptr(ws[3]), ptr(bs[3]), ptr(t1), CI, ptr(y), C, P, C, CI, 0, stream())
        torch.cuda.synchronize()
        return y, t1

    def packed(handle):
        y = torch.zeros((P, C), dtype=torch.half, device=dev)
        t1 = torch.zeros((P, CI), dtype=torch.half, device=dev)
        call(ops.dcb_nsplit_packed, handle, ptr(t2), CI, ptr(x), C, ptr(bs[0]), ptr(bs[1]), ptr(bs[2]), None, None, ptr(bs[3]),
             ptr(t1), CI, ptr(y), C, P, 0, 1, stream())
        torch.cuda.synchronize()
        return y, t1

    h = ctypes.c_void_p()
    call(ops.dcb_nsplit_pack, ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(ws[3]), C, CI, stream(), ctypes.byref(h))
    y_a, t_a = direct()
    y_h, t_h = packed(h)
    assert torch.equal(y_a, y_h) and torch.equal(t_a, t_h)
    for i, w in enumerate(ws):                   # new values at the SAME addresses
        w.copy_((_rand(tuple(w.shape), 1.0, 520 + i) / w.shape[1] ** 0.5).half())
    torch.cuda.synchronize()
    y_b, t_b = direct()
    assert not torch.equal(y_a, y_b) and not torch.equal(t_a, t_b), "the second call still used the first call's weights"
    y_h2, t_h2 = packed(h)
    assert torch.equal(y_h2, y_a) and torch.equal(t_h2, t_a), "a handle is a snapshot"
    h2 = ctypes.c_void_p()
    call(ops.dcb_nsplit_pack, ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(ws[3]), C, CI, stream(), ctypes.byref(h2))
    y_h3, t_h3 = packed(h2)
    assert torch.equal(y_h3, y_b) and torch.equal(t_h3, t_b)
    # the launch sequence on the new weights agrees (the block's semantics, not just "something changed")
    y1 = torch.zeros((P, C), dtype=torch.half, device=dev)
    t = torch.zeros((P, CI), dtype=torch.half, device=dev)
    want = torch.zeros((P, C), dtype=torch.half, device=dev)
    call(ops.conv1x1, ptr(t2), CI, ptr(ws[0]), ptr(bs[0]), ptr(x), C, None, 0, None, None, ptr(y1), C, P, CI, C, 0, stream())
    call(ops.conv1x1, ptr(y1), C, ptr(ws[1]), ptr(bs[1]), None, 0, None, 0, None, None, ptr(t), CI, P, C, 4 * CI, 3, stream())
    call(ops.conv1x1, ptr(t), CI, ptr(ws[2]), ptr(bs[2]), ptr(y1), C, None, 0, None, None, ptr(want), C, P, CI, C, 0, stream())
    torch.cuda.synchronize()
    assert torch.equal(want, y_b)
    call(ops.dcb_nsplit_free, h)
    call(ops.dcb_nsplit_free, h2)
