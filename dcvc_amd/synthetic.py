"""Seeded synthetic weights of the reference architectures and synthetic pictures.

There are no checkpoints or datasets in the reference tree or on the GPU box, so tests and
benchmarks use random-init weights of exactly the reference's shapes (dcvc_amd/arch.py). The
reference's own initialisation (xavier_normal gain 1, zero bias, q-scales 1.0,
common_model.py:49-54) lets activations grow through the ~100 residual layers and overflows fp16,
so the residual branches are damped here; q-scales and biases are made non-trivial so that every
fused epilogue and the bias folding are exercised.
"""
import math

import numpy as np
import torch


# extra per-tensor gains that bring the latents of a random network into a codec-like range
# (y of a few units, scales/means of order one, reconstruction inside [-0.5, 0.5])
_GAIN_OVERRIDES = (
    ("enc.enc_2.6.weight", 5.0),
    ("y_prior_fusion.conv.3.weight", 0.06),
    ("y_spatial_prior.conv.3.weight", 0.03),
    ("dec.dec_2.adaptor.weight", 0.0025),
    ("hyper_enc.conv.0.adaptor.weight", 0.5),
)


# the same for the inter models (spec contains q_encoder): additionally keeps the temporal recurrence
# memory -> ctx -> feature -> memory contractive so that long sequences stay inside fp16
_GAIN_OVERRIDES_INTER = (
    ("feature_adaptor_i.conv.0.adaptor.weight", 2.0),
    ("feature_adaptor_m.conv.0.adaptor.weight", 0.65),
    ("feature_extractor.conv.4.ffn.2.weight", 0.5),
    ("encoder.down.weight", 2.5),
    ("hyper_encoder.conv.0.adaptor.weight", 0.3),
    ("hyper_decoder.conv.2.adaptor.weight", 0.15),
    ("y_prior_fusion.conv.3.weight", 0.35),
    ("y_spatial_prior.conv.2.weight", 0.35),
    ("decoder.up.conv.0.weight", 0.3),
    ("decoder.conv2.weight", 0.35),
    ("recon_head.head.weight", 0.2),
)


# hierarchical inter models (HT-S / HT-L): deeper chains at width 512
_GAIN_OVERRIDES_HT = (
    ("feature_adaptor_i.conv.0.adaptor.weight", 2.0),
    ("feature_adaptor_m.conv.0.adaptor.weight", 0.4),
    ("encoder.conv1.0.adaptor.weight", 1.0),
    ("encoder.down.weight", 2.5),
    ("hyper_encoder.conv.0.dc.0.weight", 0.5),
    ("hyper_decoder.conv.2.ffn.2.weight", 0.5),
    ("y_prior_fusion.conv.3.weight", 0.35),
    ("y_spatial_prior.conv.3.weight", 0.35),
    ("decoder.up.conv.0.weight", 0.3),
    ("decoder.conv1.0.adaptor.weight", 0.5),
    ("recon_head.conv2.0.3.weight", 0.1), ("recon_head.conv2.1.3.weight", 0.1),
    ("recon_head.conv2.2.3.weight", 0.1), ("recon_head.conv2.3.3.weight", 0.1),
    ("recon_head.conv2.4.3.weight", 0.1), ("recon_head.conv2.5.3.weight", 0.1),
    ("recon_head.conv2.6.3.weight", 0.1), ("recon_head.conv2.7.3.weight", 0.1),
    ("recon_head.conv.0.5.weight", 0.1), ("recon_head.conv.1.5.weight", 0.1),
    ("recon_head.conv.2.5.weight", 0.1), ("recon_head.conv.3.5.weight", 0.1),
    ("recon_head.conv.4.5.weight", 0.1), ("recon_head.conv.5.5.weight", 0.1),
    ("recon_head.conv.6.5.weight", 0.1), ("recon_head.conv.7.5.weight", 0.1),
)


# extra damping for HT-L on top of the HT set (full-width blocks with block-level shortcuts amplify more)
_GAIN_OVERRIDES_HTL_EXTRA = (
    ("feature_adaptor_m.conv.0.adaptor.weight", 0.55),
    ("hyper_encoder.conv.1.down.weight", 0.5),
    ("hyper_encoder.conv.2.down.weight", 0.5),
    ("hyper_decoder.conv.0.up.conv.0.weight", 0.3),
    ("hyper_decoder.conv.1.up.conv.0.weight", 0.5),
    ("decoder.conv1.0.adaptor.weight", 0.6),
)


def synthetic_state_dict(spec, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    inter = "q_encoder" in spec
    ht = inter and "y_spatial_prior_reduction.weight" in spec
    overrides = _GAIN_OVERRIDES_HT if ht else _GAIN_OVERRIDES_INTER if inter else _GAIN_OVERRIDES
    if ht and "recon_head.conv.0.0.dc.0.weight" in spec:
        overrides = overrides + _GAIN_OVERRIDES_HTL_EXTRA
    for name, shape in spec.items():
        if name.startswith("bit_estimator_z."):
            # spread wide enough that table lengths vary across channels (entropy_models.py:113-149)
            std = 0.6 if name.endswith(".h") else 0.3
            t = torch.randn(shape, generator=g) * std
        elif name.startswith("q_scale"):
            qp = torch.arange(shape[0], dtype=torch.float32)[:, None]
            base = 1.0 + 0.25 * torch.rand(shape, generator=g)
            if "enc" in name:
                t = base * torch.exp((qp - 32.0) / 48.0)       # finer quantisation at high qp
            else:
                t = base * torch.exp(-(qp - 32.0) / 48.0)
        elif name in ("q_encoder", "q_decoder", "q_feature"):          # inter models
            qp = torch.arange(shape[0], dtype=torch.float32)[:, None]
            base = 1.0 + 0.25 * torch.rand(shape, generator=g)
            sign = {"q_encoder": 1.0, "q_decoder": -1.0, "q_feature": 0.5}[name]
            t = base * torch.exp(sign * (qp - 32.0) / 48.0)
        elif name.endswith(".weight"):
            cout, cin_g, kh, kw = shape
            fan_in = cin_g * kh * kw
            gain = 1.0
            if ".dc.3." in name or ".ffn.2." in name:
                gain = 0.35                                   # damp the residual branches
            elif ".ffn.0." in name:
                gain = 0.7
            elif ".dc.2." in name:
                gain = 1.2
            for key, extra in overrides:
                if name.endswith(key):
                    gain *= extra
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        elif inter and name == "y_prior_fusion.conv.3.bias":
            # (q_dec | scales | means): quantisation steps around 1, scales that straddle the skip
            # threshold, small means
            n = shape[0] // 3
            t = torch.cat([0.9 + 0.5 * torch.rand(n, generator=g),
                           0.1 + 0.6 * torch.rand(n, generator=g),
                           0.05 * torch.randn(n, generator=g)])
        elif name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * 0.02
        else:
            raise KeyError(name)
        sd[name] = t.to(dtype)
    return sd


def synthetic_frame_yuv420(height, width, index=0, seed=0):
    """8-bit YUV420 picture: low-pass filtered noise + a global pan.
    Returns (y [H, W] uint8, uv [2, H/2, W/2] uint8)."""
    rng = np.random.default_rng(seed)
    H2, W2 = height + 64, width + 96

    def smooth(h, w, cutoff):
        n = rng.standard_normal((h, w)).astype(np.float32)
        f = np.fft.rfft2(n)
        fy = np.fft.fftfreq(h)[:, None]
        fx = np.fft.rfftfreq(w)[None, :]
        f *= np.exp(-(fy * fy + fx * fx) / (2 * cutoff * cutoff))
        s = np.fft.irfft2(f, s=(h, w))
        return (s - s.mean()) / (s.std() + 1e-9)

    y_full = 128 + 45 * smooth(H2, W2, 0.03) + 12 * smooth(H2, W2, 0.2)
    u_full = 128 + 25 * smooth(H2 // 2, W2 // 2, 0.03)
    v_full = 128 + 25 * smooth(H2 // 2, W2 // 2, 0.03)
    dy, dx = (2 * index) % 64, (4 * index) % 96             # pan, even so chroma stays aligned
    y = y_full[dy:dy + height, dx:dx + width]
    u = u_full[dy // 2:dy // 2 + height // 2, dx // 2:dx // 2 + width // 2]
    v = v_full[dy // 2:dy // 2 + height // 2, dx // 2:dx // 2 + width // 2]
    to8 = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return to8(y), np.stack([to8(u), to8(v)])


def yuv420_to_x(y, uv):
    """(y, uv) uint8 -> float32 [1, 3, H, W] in [-0.5, 0.5], nearest-neighbour chroma upsampling
    (test_video.py:69-123 + transforms.py:69-80 with order=0)."""
    uv_up = np.repeat(np.repeat(uv, 2, axis=1), 2, axis=2)
    yuv = np.concatenate([y[None], uv_up], axis=0).astype(np.float32)
    return torch.from_numpy(yuv / 255.0 - 0.5).unsqueeze(0)
