"""Parameter inventory of the DCVC-UF networks (names and shapes of the reference state_dicts).

The codec itself lives in C++ (dcvc_amd/csrc/codec); Python only needs to know which tensors a
checkpoint must contain so that
  * ``DMCI`` / ``DMC`` (dcvc_amd/models.py) can hold and validate a reference checkpoint,
  * synthetic (seeded random) weights of exactly the reference architecture can be generated for
    tests and benchmarks (there are no checkpoints in the reference tree, checkpoints/.gitkeep).

Structure restated from /root/reference/src/layers/layers.py:128-188 (building blocks) and
/root/reference/src/models/image_model.py:15-148 (DMCI). Entries: name -> shape.
"""
from collections import OrderedDict

QP_NUM = 64                      # common_model.py:135-136


def _conv(spec, name, cin, cout, k=1, bias=True, groups=1):
    spec[name + ".weight"] = (cout, cin // groups, k, k)
    if bias:
        spec[name + ".bias"] = (cout,)


def depth_conv_block(spec, prefix, cin, cout, dcb2=False, force_adaptor=False):
    """layers.py:128-159 DepthConvBlock."""
    if cin != cout or force_adaptor:
        _conv(spec, prefix + "adaptor", cin, cout)
    r = 2 if dcb2 else 1
    _conv(spec, prefix + "dc.0", cout, cout // r)
    _conv(spec, prefix + "dc.2", cout // r, cout // r, k=3, groups=cout // r)
    _conv(spec, prefix + "dc.3", cout // r, cout)
    _conv(spec, prefix + "ffn.0", cout, cout * 4 // r)
    _conv(spec, prefix + "ffn.2", cout // r, cout)


def residual_block_upsample(spec, prefix, cin, cout, dcb2=False):
    """layers.py:162-173: SubpelConv2x(kernel 1, no bias) + DepthConvBlock."""
    _conv(spec, prefix + "up.conv.0", cin, cout * 4, bias=False)
    depth_conv_block(spec, prefix + "conv.", cout, cout, dcb2=dcb2)


def residual_block_stride2(spec, prefix, cin, cout, dcb2=False):
    """layers.py:176-188: pixel_unshuffle(2) + 1x1 conv + DepthConvBlock."""
    _conv(spec, prefix + "down", cin * 4, cout)
    depth_conv_block(spec, prefix + "conv.", cout, cout, dcb2=dcb2)


def bit_estimator(spec, prefix, channel):
    """entropy_models.py:78-91."""
    spec[prefix + "h"] = (QP_NUM, channel, 4)
    spec[prefix + "b"] = (QP_NUM, channel, 4)
    spec[prefix + "a"] = (QP_NUM, channel, 3)


# ---------------------------------------------------------------------------------------- DMCI
DMCI_CH_SRC, DMCI_CH_ENC_DEC, DMCI_CH_Y, DMCI_CH_Z = 192, 384, 256, 128   # image_model.py:15-18


def dmci_spec():
    s = OrderedDict()
    bit_estimator(s, "bit_estimator_z.", DMCI_CH_Z)
    c, y, z = DMCI_CH_ENC_DEC, DMCI_CH_Y, DMCI_CH_Z
    # IntraEncoder, image_model.py:50-69
    depth_conv_block(s, "enc.enc_1.", DMCI_CH_SRC, c)
    for i in range(6):
        depth_conv_block(s, "enc.enc_2.%d." % i, c, c)
    _conv(s, "enc.enc_2.6", c, y, k=3)
    # IntraHyperEncoder, image_model.py:86-95
    depth_conv_block(s, "hyper_enc.conv.0.", y, z)
    residual_block_stride2(s, "hyper_enc.conv.1.", z, z)
    residual_block_stride2(s, "hyper_enc.conv.2.", z, z)
    # IntraHyperDecoder, image_model.py:72-83
    residual_block_upsample(s, "hyper_dec.conv.0.", z, z)
    residual_block_upsample(s, "hyper_dec.conv.1.", z, z)
    depth_conv_block(s, "hyper_dec.conv.2.", z, y)
    # IntraYPriorFusion, image_model.py:112-123
    depth_conv_block(s, "y_prior_fusion.conv.0.", y, 2 * y)
    depth_conv_block(s, "y_prior_fusion.conv.1.", 2 * y, 2 * y)
    depth_conv_block(s, "y_prior_fusion.conv.2.", 2 * y, 2 * y)
    _conv(s, "y_prior_fusion.conv.3", 2 * y, 2 * y)
    # image_model.py:136-140
    _conv(s, "y_spatial_prior_reduction", 2 * y, y)
    for i in (1, 2, 3):
        depth_conv_block(s, "y_spatial_prior_adaptor_%d." % i, 2 * y, 2 * y, force_adaptor=True)
    depth_conv_block(s, "y_spatial_prior.conv.0.", 2 * y, 2 * y)
    depth_conv_block(s, "y_spatial_prior.conv.1.", 2 * y, 2 * y)
    depth_conv_block(s, "y_spatial_prior.conv.2.", 2 * y, 2 * y)
    _conv(s, "y_spatial_prior.conv.3", 2 * y, 2 * y)
    # IntraDecoder, image_model.py:21-47
    residual_block_upsample(s, "dec.dec_1.0.", y, c)
    for i in range(1, 13):
        depth_conv_block(s, "dec.dec_1.%d." % i, c, c)
    depth_conv_block(s, "dec.dec_2.", c, DMCI_CH_SRC)
    # image_model.py:144-147
    s["q_scale_enc"] = (QP_NUM, c)
    s["q_scale_dec"] = (QP_NUM, c)
    s["q_scale_y_enc"] = (QP_NUM, y)
    s["q_scale_y_dec"] = (QP_NUM, y)
    return s


def param_count(spec):
    n = 0
    for shape in spec.values():
        k = 1
        for d in shape:
            k *= d
        n += k
    return n


# ---------------------------------------------------------------------------------------- DMC LD
# /root/reference/src/models/video_model_ld.py:16-21
LD_FRAME_DELAY = 1
LD_CH_SRC, LD_CH_Y, LD_CH_Z, LD_CH_D, LD_CH_M = 192, 128, 128, 256, 256


def dmc_ld_spec():
    """Low-delay inter model (video_model_ld.py:24-230), every DepthConvBlock is dcb2."""
    s = OrderedDict()
    bit_estimator(s, "bit_estimator_z.", LD_CH_Z)
    y, z, d, m = LD_CH_Y, LD_CH_Z, LD_CH_D, LD_CH_M

    def chain(prefix, cin, c, n):
        depth_conv_block(s, prefix + "0.", cin, c, dcb2=True)
        for i in range(1, n):
            depth_conv_block(s, prefix + "%d." % i, c, c, dcb2=True)

    chain("feature_adaptor_i.conv.", LD_CH_SRC, m, 4)           # :63-77
    chain("feature_adaptor_m.conv.", m + d, m, 4)               # :80-92
    chain("feature_extractor.conv.", m, m, 5)                   # :95-110
    chain("encoder.conv1.", LD_CH_SRC + m, d, 2)                # :43-60
    depth_conv_block(s, "encoder.conv2.", d, d, dcb2=True)
    _conv(s, "encoder.down", d, y, k=3)
    depth_conv_block(s, "hyper_encoder.conv.0.", y, z, dcb2=True)            # :128-139
    residual_block_stride2(s, "hyper_encoder.conv.1.", z, z, dcb2=True)
    residual_block_stride2(s, "hyper_encoder.conv.2.", z, z, dcb2=True)
    residual_block_upsample(s, "hyper_decoder.conv.0.", z, z, dcb2=True)     # :113-125
    residual_block_upsample(s, "hyper_decoder.conv.1.", z, z, dcb2=True)
    depth_conv_block(s, "hyper_decoder.conv.2.", z, y, dcb2=True)
    residual_block_stride2(s, "temporal_prior_encoder.conv.", m, 2 * y, dcb2=True)   # :187-194
    chain("y_prior_fusion.conv.", 3 * y, 3 * y, 3)                           # :142-154
    _conv(s, "y_prior_fusion.conv.3", 3 * y, 3 * y)
    depth_conv_block(s, "y_spatial_prior.conv.0.", 4 * y, 2 * y, dcb2=True)  # :174-184
    depth_conv_block(s, "y_spatial_prior.conv.1.", 2 * y, 2 * y, dcb2=True)
    _conv(s, "y_spatial_prior.conv.2", 2 * y, y)
    _conv(s, "decoder.up.conv.0", y, 4 * d, bias=False)                      # :24-40
    chain("decoder.conv1.", d + m, d, 3)
    _conv(s, "decoder.conv2", d, d)
    chain("recon_head.conv.", d, d, 3)                                       # :157-171
    _conv(s, "recon_head.head", d, LD_CH_SRC)
    s["q_encoder"] = (QP_NUM, d)                                             # :216-218
    s["q_decoder"] = (QP_NUM, d)
    s["q_feature"] = (QP_NUM, 2 * y)
    return s


# ---------------------------------------------------------------------------------------- DMC HT-S / HT-L
# /root/reference/src/models/video_model_ht.py:16-23
HT_FRAME_DELAY = 8
HT_CH_SRC_INTRA = 192
HT_CH_SRC = HT_CH_SRC_INTRA * HT_FRAME_DELAY
HT_CH_Y, HT_CH_Z, HT_CH_D, HT_CH_M, HT_CH_RECON = 256, 128, 512, 512, 256


def dmc_ht_spec(is_hts=True):
    """Hierarchical inter models, 8 pictures per call (video_model_ht.py:26-355). HT-S uses the
    half-width blocks (dcb2) in the picture-resolution networks, HT-L full-width ones and deeper
    chains, a 3x3 biased sub-pixel upsampler in the decoder and scales from the spatial prior."""
    s = OrderedDict()
    bit_estimator(s, "bit_estimator_z.", HT_CH_Z)
    y, z, d, m, r = HT_CH_Y, HT_CH_Z, HT_CH_D, HT_CH_M, HT_CH_RECON
    half = is_hts

    def chain(prefix, cin, c, n, dcb2):
        depth_conv_block(s, prefix + "0.", cin, c, dcb2=dcb2)
        for i in range(1, n):
            depth_conv_block(s, prefix + "%d." % i, c, c, dcb2=dcb2)

    chain("feature_adaptor_i.conv.", HT_CH_SRC_INTRA, m, 4 if is_hts else 3, half)     # :95-114
    chain("feature_adaptor_m.conv.", m + d, m, 6 if is_hts else 10, half)              # :117-146
    chain("feature_extractor.conv.", m, d, 5 if is_hts else 2, half)                   # :149-166
    chain("encoder.conv1.", HT_CH_SRC + d, d, 6 if is_hts else 7, half)                # :63-92
    _conv(s, "encoder.down", d, y, k=3)
    depth_conv_block(s, "hyper_encoder.conv.0.", y, y)                                 # :186-201
    residual_block_stride2(s, "hyper_encoder.conv.1.", y, y)
    residual_block_stride2(s, "hyper_encoder.conv.2.", y, z)
    for i, (cin, cout) in enumerate(((z, y), (y, y))):                                 # :169-183
        _conv(s, "hyper_decoder.conv.%d.up.conv.0" % i, cin, cout * 4, bias=not is_hts)
        depth_conv_block(s, "hyper_decoder.conv.%d.conv." % i, cout, cout)
    depth_conv_block(s, "hyper_decoder.conv.2.", y, y)
    residual_block_stride2(s, "temporal_prior_encoder.conv.", d, 2 * y)                # :305-316
    chain("y_prior_fusion.conv.", 3 * y, 3 * y, 3, False)                              # :204-215
    _conv(s, "y_prior_fusion.conv.3", 3 * y, 3 * y)
    _conv(s, "y_spatial_prior_reduction", 3 * y, y)                                    # :337-344
    for i in (1, 2, 3):
        depth_conv_block(s, "y_spatial_prior_adaptor_%d." % i, 2 * y, 2 * y, force_adaptor=True)
    chain("y_spatial_prior.conv.", 2 * y, 2 * y, 3, False)                             # :278-290
    _conv(s, "y_spatial_prior.conv.3", 2 * y, y if is_hts else 2 * y)
    if is_hts:                                                                         # :26-60
        _conv(s, "decoder.up.conv.0", y, 4 * d, bias=False)
    else:
        _conv(s, "decoder.up.conv.0", y, 4 * d, k=3)
    chain("decoder.conv1.", 2 * d, d, 7 if is_hts else 11, half)
    if is_hts:                                                                         # :218-275
        for i in range(HT_FRAME_DELAY // 2):
            depth_conv_block(s, "recon_head.conv1.%d.0." % i, d, d)
        for i in range(HT_FRAME_DELAY):
            chain("recon_head.conv2.%d." % i, d, r, 3, False)
            _conv(s, "recon_head.conv2.%d.3" % i, r, HT_CH_SRC_INTRA)
    else:
        for i in range(HT_FRAME_DELAY):
            chain("recon_head.conv.%d." % i, d, r, 5, False)
            _conv(s, "recon_head.conv.%d.5" % i, r, HT_CH_SRC_INTRA)
    s["q_encoder"] = (QP_NUM, d)                                                       # :349-351
    s["q_decoder"] = (QP_NUM, d)
    s["q_feature"] = (QP_NUM, d)
    return s
