"""fp32 PyTorch restatement of the reference's CPU-runnable path (TEST / BASELINE INFRASTRUCTURE ONLY).

The reference has no CPU compress(); what it can run on host cores is the training graph
`DMCI.forward_one_frame(x, qp, recon_only=True)` in fp32 (/root/reference/src/models/image_model.py:
150-171; SURVEY fact 1, section 8d "Reference CPU path timed beside it"). /root/reference does not
exist on the GPU box, so bench.py's `cpu_baseline` leg times THIS restatement there: plain
torch.nn.functional ops on the same state_dict (dcvc_amd/arch.py names), same dataflow:

  IntraEncoder / IntraHyperEncoder / IntraHyperDecoder / IntraYPriorFusion / IntraSpatialPrior /
  IntraDecoder            image_model.py:21-123
  DepthConvBlock, ResidualBlockUpsample, ResidualBlockWithStride2, WSiLU(ChunkAdd)   layers.py:92-188
  forward_prior_4x, process_with_mask, get_mask_4x        common_model.py:123-132, 174-195, 231-282

tests/test_oracle_cpu.py checks it against the reference module itself (when /root/reference is
present) and against the golden x_hat made from the reference (tests/golden/dmci_golden.npz).
"""
import torch
import torch.nn.functional as F


def _conv1x1(x, sd, p):
    return F.conv2d(x, sd[p + "weight"], sd.get(p + "bias"))


def _wsilu(x):
    return torch.sigmoid(4.0 * x) * x


def dcb(x, sd, p, shortcut=False):
    """layers.py:128-159"""
    if p + "adaptor.weight" in sd:
        x = _conv1x1(x, sd, p + "adaptor.")
    t = _wsilu(_conv1x1(x, sd, p + "dc.0."))
    w = sd[p + "dc.2.weight"]
    t = F.conv2d(t, w, sd[p + "dc.2.bias"], padding=1, groups=w.shape[0])
    out = _conv1x1(t, sd, p + "dc.3.") + x
    t = _wsilu(_conv1x1(out, sd, p + "ffn.0."))
    t = t[:, 0::4] + t[:, 1::4] + t[:, 2::4] + t[:, 3::4]
    out = _conv1x1(t, sd, p + "ffn.2.") + out
    return out + x if shortcut else out


def rb_upsample(x, sd, p):
    """layers.py:162-173 (SubpelConv2x kernel 1 + DepthConvBlock with shortcut)"""
    out = F.pixel_shuffle(F.conv2d(x, sd[p + "up.conv.0.weight"], sd.get(p + "up.conv.0.bias")), 2)
    return dcb(out, sd, p + "conv.", shortcut=True)


def rb_stride2(x, sd, p):
    """layers.py:176-188"""
    out = _conv1x1(F.pixel_unshuffle(x, 2), sd, p + "down.")
    return dcb(out, sd, p + "conv.", shortcut=True)


def _mask_4x(C, H, W, device):
    """common_model.py:174-195"""
    def one(pattern):
        m = torch.tensor(pattern, dtype=torch.bool, device=device)
        return m.repeat((H + 1) // 2, (W + 1) // 2)[:H, :W]
    m0, m1, m2, m3 = one(((1, 0), (0, 0))), one(((0, 1), (0, 0))), one(((0, 0), (1, 0))), one(((0, 0), (0, 1)))
    q = C // 4
    ones = torch.ones((1, q, H, W), dtype=torch.bool, device=device)
    cat = lambda *ms: torch.cat([ones * m for m in ms], dim=1)
    return cat(m0, m1, m2, m3), cat(m3, m2, m1, m0), cat(m2, m3, m0, m1), cat(m1, m0, m3, m2)


def _process_with_mask(y, scales, means, mask):
    """common_model.py:123-132 (no clamp, no skip mode, torch.round = half to even)"""
    means_hat = means * mask
    y_q = torch.round((y - means_hat) * mask)
    return y_q + means_hat


@torch.no_grad()
def forward_one_frame(sd, x, qp):
    """x: [1, 3, H, W] fp32 in [-0.5, 0.5], H and W multiples of 64 -> x_hat [1, 3, H, W]."""
    q_enc = sd["q_scale_enc"][qp][None, :, None, None]
    q_dec = sd["q_scale_dec"][qp][None, :, None, None]
    q_y_enc = sd["q_scale_y_enc"][qp][None, :, None, None]
    q_y_dec = sd["q_scale_y_dec"][qp][None, :, None, None]
    # IntraEncoder
    out = dcb(F.pixel_unshuffle(x, 8), sd, "enc.enc_1.") * q_enc
    for i in range(6):
        out = dcb(out, sd, "enc.enc_2.%d." % i)
    y = F.conv2d(out, sd["enc.enc_2.6.weight"], sd["enc.enc_2.6.bias"], stride=2, padding=1)
    # hyper codec
    z = rb_stride2(rb_stride2(dcb(y, sd, "hyper_enc.conv.0."), sd, "hyper_enc.conv.1."), sd, "hyper_enc.conv.2.")
    z_hat = torch.round(z)
    params = dcb(rb_upsample(rb_upsample(z_hat, sd, "hyper_dec.conv.0."), sd, "hyper_dec.conv.1."), sd, "hyper_dec.conv.2.")
    for i in range(3):
        params = dcb(params, sd, "y_prior_fusion.conv.%d." % i)
    params = _conv1x1(params, sd, "y_prior_fusion.conv.3.")[:, :, :y.shape[2], :y.shape[3]]
    # forward_prior_4x
    C = y.shape[1]
    scales, means = params.chunk(2, 1)
    y = y * q_y_enc
    common = _conv1x1(params, sd, "y_spatial_prior_reduction.")
    masks = _mask_4x(C, y.shape[2], y.shape[3], y.device)
    y_hat = _process_with_mask(y, scales, means, masks[0])
    for k in range(1, 4):
        t = dcb(torch.cat((y_hat, common), dim=1), sd, "y_spatial_prior_adaptor_%d." % k)
        for i in range(3):
            t = dcb(t, sd, "y_spatial_prior.conv.%d." % i)
        scales, means = _conv1x1(t, sd, "y_spatial_prior.conv.3.").chunk(2, 1)
        y_hat = y_hat + _process_with_mask(y, scales, means, masks[k])
    y_hat = y_hat * q_y_dec
    # IntraDecoder
    out = rb_upsample(y_hat, sd, "dec.dec_1.0.")
    for i in range(1, 13):
        out = dcb(out, sd, "dec.dec_1.%d." % i)
    out = dcb(out * q_dec, sd, "dec.dec_2.")
    return F.pixel_shuffle(out, 8)


def time_forward(sd, height, width, qp, threads, repeats=1, x=None, keep=None):
    """seconds per picture of forward_one_frame on `threads` host threads (fp32, channels_last). `x`: the picture to
    reconstruct ([1, 3, height, width] fp32; default seeded uniform noise); `keep`: a list that receives the last x_hat."""
    import time
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        if x is None:
            g = torch.Generator().manual_seed(0)
            x = torch.rand((1, 3, height, width), generator=g) - 0.5
        x = x.contiguous(memory_format=torch.channels_last)
        sd32 = {k: (v.float().contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v.float())
                for k, v in sd.items() if hasattr(v, "dim")}
        forward_one_frame(sd32, x[:, :, :64, :64], qp)          # warm-up (thread pool, kernels)
        t0 = time.perf_counter()
        for _ in range(repeats):
            out = forward_one_frame(sd32, x, qp)
        if keep is not None:
            keep.append(out)
        return (time.perf_counter() - t0) / repeats
    finally:
        torch.set_num_threads(prev)
