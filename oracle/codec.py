"""CPU oracle of the DCVC-UF *inference* path (TEST INFRASTRUCTURE ONLY - never imported by the
product; see oracle/__init__.py).

Restates, op by op, what the reference's native proxies do (there is no CPU implementation of
compress()/decompress() in the reference), under the MI355X build's arithmetic
policy: fp16 NHWC tensors wherever the proxy materialises one, fp32 math inside an op,
contractions accumulated exactly like the gfx950 matrix core (oracle/nn_oracle.c).

  DepthConvBlock          layers_proxy.cpp:71-101   (forward), :160-206 (weight folding)
  ResidualBlockWithStride2  layers_proxy.cpp:234-267 (== layers.py:176-188)
  ResidualBlockUpsample / SubpelConv2x  layers_proxy.cpp:208-232, 269-324 (== layers.py:92-173)
  DMCI networks           dmci_proxy.cpp:14-294     (== image_model.py:21-123)
  DMCI compress           dmci_proxy.cpp:296-421, entropy calls :818-845
  DMCI decompress         dmci_proxy.cpp:423-602, worker :846-871
  padding / ec_parallel   dmc_common.cpp:31-35, 64-83

Tensors: numpy float16 [H, W, C]. Weights: dict name -> numpy array in PyTorch layout.
"""
import numpy as np

from oracle import nn
from oracle import rans as orc_rans
from oracle import symbols_np as sym

F16 = np.float16
MIN_SYMBOLS_PER_STREAM = 32768      # def_const.h:18
MAX_EC_PARALLEL = 8                 # py_rans.h:15


def to_np_state_dict(state_dict):
    """torch / numpy state_dict -> numpy; floating tensors become fp16 (finalize_model:
    net.half(), test_video.py:27-29)."""
    out = {}
    for k, v in state_dict.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if a.dtype.kind == "f":
            a = a.astype(F16)
        out[k] = a
    return out


def compute_ec_parallel(symbol_count):
    return max(1, min(MAX_EC_PARALLEL, symbol_count // MIN_SYMBOLS_PER_STREAM))


def get_padding_size(h, w, p):
    nh = (h + p - 1) // p * p
    nw = (w + p - 1) // p * p
    return nw - w, nh - h        # right, bottom


def replicate_pad(x, pad_b, pad_r):
    if pad_b == 0 and pad_r == 0:
        return x
    return np.pad(x, ((0, pad_b), (0, pad_r), (0, 0)), mode="edge")


class Net:
    """Functional building blocks over a numpy state_dict."""

    def __init__(self, sd):
        self.sd = sd
        self._folded = {}

    def has(self, name):
        return name in self.sd

    def w(self, name):
        return self.sd[name]

    def conv1x1(self, x, prefix, **kw):
        return nn.conv1x1(x, self.sd[prefix + "weight"], self.sd.get(prefix + "bias"), **kw)

    def dcb(self, x, p, shortcut=False, q=None, q2=None):
        """DepthConvBlockProxy::forward, layers_proxy.cpp:71-101."""
        sd = self.sd
        out = x
        if p + "adaptor.weight" in sd:
            out = self.conv1x1(x, p + "adaptor.")
        sc = out
        t = self.conv1x1(out, p + "dc.0.", wsilu=True)
        t = nn.dwconv3x3(t, sd[p + "dc.2.weight"])
        if p not in self._folded:
            self._folded[p] = nn.fold_dw_bias(sd[p + "dc.3.weight"], sd[p + "dc.2.bias"], sd[p + "dc.3.bias"])
        out = nn.conv1x1(t, sd[p + "dc.3.weight"], self._folded[p], r1=sc)
        sc_ffn = out
        t = self.conv1x1(out, p + "ffn.0.", wsilu=True, chunk_add=True)
        return nn.conv1x1(t, sd[p + "ffn.2.weight"], sd[p + "ffn.2.bias"], r1=sc_ffn,
                          r2=sc if shortcut else None, q=q, q2=q2)

    def rb_stride2(self, x, p, shortcut=True):
        """pixel_unshuffle(2) + 1x1 (== the folded 2x2 stride-2 conv) + DCB."""
        w = self.sd[p + "down.weight"]                      # [C', 4C, 1, 1], ch = c*4 + dy*2 + dx
        cout, c4 = w.shape[0], w.shape[1]
        w2 = w.reshape(cout, c4 // 4, 2, 2)                 # PyTorch conv layout [C', C, ky, kx]
        out = nn.conv_kxk(x, w2, self.sd[p + "down.bias"], 2, 2, 0)
        return self.dcb(out, p + "conv.", shortcut=shortcut)

    def subpel(self, x, p):
        """SubpelConv2xProxy::forward (layers_proxy.cpp:269-276): without bias a 2x2 stride-2
        transposed conv; with bias a k x k conv + bias (rounded to fp16) + pixel_shuffle(2)."""
        w = self.sd[p + "conv.0.weight"]
        if p + "conv.0.bias" not in self.sd:
            return nn.subpel_conv1x1(x, w)
        k = w.shape[-1]
        out = nn.conv_kxk(x, w, self.sd[p + "conv.0.bias"], k, 1, k // 2)
        return nn.pixel_shuffle(out, 2)

    def rb_upsample(self, x, p, shortcut=True):
        return self.dcb(self.subpel(x, p + "up."), p + "conv.", shortcut=shortcut)

    def chain(self, x, prefix, q_last=None):
        """nn.Sequential of DepthConvBlocks prefix0., prefix1., ...; q_last: scale fused into the
        last block (DepthConvBlockProxy::forward(x, quant), layers_proxy.cpp:92-95)."""
        n = 0
        while prefix + "%d.dc.0.weight" % n in self.sd:
            n += 1
        assert n > 0, prefix
        for i in range(n):
            x = self.dcb(x, prefix + "%d." % i, q=q_last if i == n - 1 else None)
        return x


class DMCIOracle:
    """CPU restatement of DMCIProxy (set_param / compress / decompress)."""
    CH_Y, CH_Z = 256, 128

    def __init__(self, state_dict, skip_thres, cdf_tables=None):
        self.sd = to_np_state_dict(state_dict)
        self.net = Net(self.sd)
        self.skip_thres = float(skip_thres)
        self.tables = orc_rans.Tables()
        if cdf_tables is not None:
            self.set_cdf(*cdf_tables)
        self.debug = {}

    def set_cdf(self, z_cdf, z_len, y_cdf, y_len):
        self.tables.set_cdf(z_cdf, z_len, 0)
        self.tables.set_cdf(y_cdf, y_len, 1)

    # ---- sub-networks (dmci_proxy.cpp:14-294)
    def encoder(self, x_unshuffled, qp):
        n = self.net
        out = n.dcb(x_unshuffled, "enc.enc_1.", q2=self.sd["q_scale_enc"][qp])
        for i in range(6):
            out = n.dcb(out, "enc.enc_2.%d." % i)
        return nn.conv_kxk(out, self.sd["enc.enc_2.6.weight"], self.sd["enc.enc_2.6.bias"], 3, 2, 1)

    def hyper_encoder(self, y_pad):
        n = self.net
        out = n.dcb(y_pad, "hyper_enc.conv.0.")
        out = n.rb_stride2(out, "hyper_enc.conv.1.")
        return n.rb_stride2(out, "hyper_enc.conv.2.")

    def hyper_decoder(self, z_hat):
        n = self.net
        out = n.rb_upsample(z_hat, "hyper_dec.conv.0.")
        out = n.rb_upsample(out, "hyper_dec.conv.1.")
        return n.dcb(out, "hyper_dec.conv.2.")

    def prior_fusion(self, hyper_params):
        n = self.net
        out = n.dcb(hyper_params, "y_prior_fusion.conv.0.")
        out = n.dcb(out, "y_prior_fusion.conv.1.")
        out = n.dcb(out, "y_prior_fusion.conv.2.")
        return n.conv1x1(out, "y_prior_fusion.conv.3.")

    def spatial_prior(self, y_hat_so_far, reduced, k):
        n = self.net
        cat = np.concatenate([y_hat_so_far, reduced], axis=-1)
        out = n.dcb(cat, "y_spatial_prior_adaptor_%d." % k)
        out = n.dcb(out, "y_spatial_prior.conv.0.")
        out = n.dcb(out, "y_spatial_prior.conv.1.")
        out = n.dcb(out, "y_spatial_prior.conv.2.")
        return n.conv1x1(out, "y_spatial_prior.conv.3.")

    def decoder(self, y_hat, qp):
        n = self.net
        out = n.rb_upsample(y_hat, "dec.dec_1.0.")
        for i in range(1, 12):
            out = n.dcb(out, "dec.dec_1.%d." % i)
        out = n.dcb(out, "dec.dec_1.12.", q2=self.sd["q_scale_dec"][qp])
        out = n.dcb(out, "dec.dec_2.")
        out = nn.pixel_shuffle(out, 8)
        return np.clip(out, F16(-0.5), F16(0.5)).astype(F16)          # shuffle.cu:53-56

    def _priors(self, z_hat, yH, yW):
        hyper = self.hyper_decoder(z_hat)
        params = self.prior_fusion(hyper)[:yH, :yW]                    # crop_hyper_params
        reduced = self.net.conv1x1(params, "y_spatial_prior_reduction.")
        return params, reduced

    # ---- DMCIProxy::compress (dmci_proxy.cpp:296-421)
    def compress(self, x, qp, padding_b=None, padding_r=None):
        """x: float16 [H, W, 3] in [-0.5, 0.5] (unpadded). Returns dict(bit_stream, x_hat
        [H16, W16, 3] fp16, ec_parallel)."""
        H, W, _ = x.shape
        pr, pb = get_padding_size(H, W, 16)
        if padding_b is not None:
            assert (padding_b, padding_r) == (pb, pr)
        xu = nn.pixel_unshuffle(replicate_pad(x.astype(F16), pb, pr), 8)       # cat_and_pad.cu:7-31
        y = self.encoder(xu, qp)
        yH, yW, C = y.shape
        pr4, pb4 = get_padding_size(yH, yW, 4)
        z = self.hyper_encoder(replicate_pad(y, pb4, pr4))
        z_hat, z_i8 = sym.round_z(z)
        params, reduced = self._priors(z_hat, yH, yW)
        scales, means = params[..., :C], params[..., C:]
        y = nn.mul_channel(y, self.sd["q_scale_y_enc"][qp])
        masks = sym.get_mask_4x(yH, yW, C)
        y_hat_so_far = None
        y_syms = []
        self.debug = dict(y=y, z_i8=z_i8, scales=[], means=[])
        for k in range(4):
            self.debug["scales"].append(scales)
            self.debug["means"].append(means)
            y_q, y_hat, s_hat = sym.process_with_mask(y, scales, means, masks[k], self.skip_thres)
            comb, keep = sym.build_index_enc(sym.fold4(y_q), sym.fold4(s_hat), self.skip_thres)
            y_syms.append(comb[keep])
            y_hat_so_far = y_hat if y_hat_so_far is None else (y_hat_so_far + y_hat).astype(F16)
            if k < 3:
                sp = self.spatial_prior(y_hat_so_far, reduced, k + 1)
                scales, means = sp[..., :C], sp[..., C:]
        y_hat = nn.mul_channel(y_hat_so_far, self.sd["q_scale_y_dec"][qp])
        self.debug["y_hat"] = y_hat
        # worker(): entropy coding, groups 3,2,1,0 then z (dmci_proxy.cpp:818-845)
        total = sum(len(s) for s in y_syms)
        ec = compute_ec_parallel(total)
        segs = [("y", y_syms[k]) for k in (3, 2, 1, 0)]
        segs.append(("z", z_i8.reshape(-1), qp * self.CH_Z, self.CH_Z))
        stream = orc_rans.encode(self.tables, segs, ec)
        x_hat = self.decoder(y_hat, qp)
        self.debug["y_syms"] = y_syms
        return dict(bit_stream=stream.tobytes(), x_hat=x_hat, ec_parallel=ec)

    # ---- DMCIProxy::decompress (dmci_proxy.cpp:423-602)
    def decompress(self, bit_stream, qp, height, width, ec_parallel):
        C = self.CH_Y
        zH, zW = (height + 63) // 64, (width + 63) // 64
        yH, yW = (height + 15) // 16, (width + 15) // 16
        dec = orc_rans.Decoder(self.tables, np.frombuffer(bit_stream, dtype=np.uint8), ec_parallel)
        z_i8 = dec.decode_z(self.CH_Z * zH * zW, qp * self.CH_Z, self.CH_Z).reshape(zH, zW, self.CH_Z)
        z_hat = z_i8.astype(F16)
        params, reduced = self._priors(z_hat, yH, yW)
        scales, means = params[..., :C], params[..., C:]
        masks = sym.get_mask_4x(yH, yW, C)
        y_hat_so_far = None
        for k in range(4):
            s_r = sym.fold4(np.where(masks[k], scales, F16(0)).astype(F16))
            idx, keep = sym.build_index_dec(s_r, self.skip_thres)
            decoded = dec.decode_y(idx[keep])
            y_q_r = sym.recover(decoded, keep, (yH, yW, C // 4))
            y_hat = sym.restore_y_4x(y_q_r, means, masks[k])
            y_hat_so_far = y_hat if y_hat_so_far is None else (y_hat_so_far + y_hat).astype(F16)
            if k < 3:
                sp = self.spatial_prior(y_hat_so_far, reduced, k + 1)
                scales, means = sp[..., :C], sp[..., C:]
        dec.close()
        y_hat = nn.mul_channel(y_hat_so_far, self.sd["q_scale_y_dec"][qp])
        return self.decoder(y_hat, qp)


class DMCLDOracle:
    """CPU restatement of DMCLDProxy (dmc_ld_proxy.cpp:407-593): low-delay inter codec, one
    picture per call, temporal state = (feature_i | memory, feature_p) + ctx + temporal params.
    Networks: video_model_ld.py:24-194 (every DepthConvBlock is dcb2)."""
    CH_SRC, CH_Y, CH_Z, CH_D, CH_M = 192, 128, 128, 256, 256

    def __init__(self, state_dict, skip_thres, cdf_tables=None):
        self.sd = to_np_state_dict(state_dict)
        self.net = Net(self.sd)
        self.skip_thres = float(skip_thres)
        self.tables = orc_rans.Tables()
        if cdf_tables is not None:
            self.tables.set_cdf(cdf_tables[0], cdf_tables[1], 0)
            self.tables.set_cdf(cdf_tables[2], cdf_tables[3], 1)
        self.clear()
        self.debug = {}

    def clear(self):
        self.feature_i = None        # recon-head output before the shuffle / unshuffled I picture
        self.memory = None
        self.feature_p = None
        self.ctx = None
        self.temporal = None
        self.memory_has_value = False

    # ---- sub-networks
    def _chain(self, x, prefix, n, **last_kw):
        for i in range(n):
            x = self.net.dcb(x, prefix + "%d." % i, **(last_kw if i == n - 1 else {}))
        return x

    def fa_i(self, x):
        return self._chain(x, "feature_adaptor_i.conv.", 4)

    def fa_m(self, memory, feature):
        return self._chain(np.concatenate([memory, feature], axis=-1), "feature_adaptor_m.conv.", 4)

    def fe(self, memory):
        return self._chain(memory, "feature_extractor.conv.", 5)

    def tpe(self, memory):
        return self.net.rb_stride2(memory, "temporal_prior_encoder.conv.", shortcut=False)

    def encoder(self, xu, ctx, qp):
        n = self.net
        out = self._chain(np.concatenate([xu, ctx], axis=-1), "encoder.conv1.", 2)
        out = n.dcb(out, "encoder.conv2.", q=self.sd["q_encoder"][qp])     # ..._shortcut_with_quant
        return nn.conv_kxk(out, self.sd["encoder.down.weight"], self.sd["encoder.down.bias"], 3, 2, 1)

    def hyper_encoder(self, y_pad):
        n = self.net
        out = n.dcb(y_pad, "hyper_encoder.conv.0.")
        out = n.rb_stride2(out, "hyper_encoder.conv.1.", shortcut=False)
        return n.rb_stride2(out, "hyper_encoder.conv.2.", shortcut=False)

    def hyper_decoder(self, z_hat):
        n = self.net
        out = n.rb_upsample(z_hat, "hyper_decoder.conv.0.", shortcut=False)
        out = n.rb_upsample(out, "hyper_decoder.conv.1.", shortcut=False)
        return n.dcb(out, "hyper_decoder.conv.2.")

    def prior_fusion(self, hyper, temporal_q):
        out = self._chain(np.concatenate([hyper, temporal_q], axis=-1), "y_prior_fusion.conv.", 3)
        return self.net.conv1x1(out, "y_prior_fusion.conv.3.")

    def spatial_prior(self, y_hat, common):
        out = self._chain(np.concatenate([y_hat, common], axis=-1), "y_spatial_prior.conv.", 2)
        return self.net.conv1x1(out, "y_spatial_prior.conv.2.")

    def decoder(self, y_hat, ctx, qp):
        up = nn.subpel_conv1x1(y_hat, self.sd["decoder.up.conv.0.weight"])
        out = self._chain(np.concatenate([up, ctx], axis=-1), "decoder.conv1.", 3)
        return nn.conv1x1(out, self.sd["decoder.conv2.weight"], self.sd["decoder.conv2.bias"],
                          q=self.sd["q_decoder"][qp])                        # conv1x1_bias_with_quant

    def recon_head(self, feature):
        out = self._chain(feature, "recon_head.conv.", 3)
        head = self.net.conv1x1(out, "recon_head.head.")
        x_hat = np.clip(nn.pixel_shuffle(head, 8), F16(-0.5), F16(0.5)).astype(F16)
        return head, x_hat

    # ---- DMCLDProxy::add_ref_feature_from_frame (dmc_ld_proxy.cpp:407-418)
    def add_ref_feature_from_frame(self, frame, apply_adaptor):
        """frame: the I codec's reconstruction [H16, W16, 3] fp16."""
        self.feature_i = nn.pixel_unshuffle(frame.astype(F16), 8)
        if apply_adaptor:
            self.memory = self.fa_i(self.feature_i)
            self.ctx = self.fe(self.memory)
            self.temporal = self.tpe(self.memory)
        self.memory_has_value = bool(apply_adaptor)

    def _priors(self, z_hat, qp):
        hyper = self.hyper_decoder(z_hat)
        hyper = hyper[:self.temporal.shape[0], :self.temporal.shape[1]]
        temporal_q = nn.mul_channel(self.temporal, self.sd["q_feature"][qp])
        common = self.prior_fusion(hyper, temporal_q)
        C = self.CH_Y
        return common, common[..., :C], common[..., C:2 * C], common[..., 2 * C:]

    # ---- DMCLDProxy::compress (dmc_ld_proxy.cpp:420-473)
    def compress(self, x, qp, reset_feature_memory):
        H, W, _ = x.shape
        pr, pb = get_padding_size(H, W, 16)
        xu = nn.pixel_unshuffle(replicate_pad(x.astype(F16), pb, pr), 8)
        y = self.encoder(xu, self.ctx, qp)
        yH, yW, C = y.shape
        pr4, pb4 = get_padding_size(yH, yW, 4)
        z = self.hyper_encoder(replicate_pad(y, pb4, pr4))
        z_hat, z_i8 = sym.round_z(z)
        common, q_dec, scales, means = self._priors(z_hat, qp)
        y = sym.divide_with_clamp(y, q_dec)
        mask_0, mask_1 = sym.get_mask_2x(yH, yW, C)
        y_q0, y_hat0 = sym.process_with_mask_2x(y, scales, means, mask_0, self.skip_thres)
        means1 = self.spatial_prior(y_hat0, common)
        y_q1, y_hat1 = sym.process_with_mask_2x(y, scales, means1, mask_1, self.skip_thres)
        y_q = (y_q0 + y_q1).astype(F16)
        y_hat = ((y_hat0 + y_hat1).astype(F16) * sym.clamp_min_half(q_dec)).astype(F16)
        comb, keep = sym.build_index_enc(y_q, scales, self.skip_thres)
        y_sym = comb[keep]
        ec = compute_ec_parallel(len(y_sym))
        stream = orc_rans.encode(self.tables, [("y", y_sym), ("z", z_i8.reshape(-1), qp * self.CH_Z, self.CH_Z)], ec)
        self.debug = dict(y=y, y_hat=y_hat, z_i8=z_i8, y_sym=y_sym)
        # lambda_enc_1: decoder, state update
        self.feature_p = self.decoder(y_hat, self.ctx, qp)
        if reset_feature_memory:
            head, _ = self.recon_head(self.feature_p)
            self.memory = self.fa_i(head)
        else:
            self.memory = self.fa_m(self.memory, self.feature_p)
        self.ctx = self.fe(self.memory)
        self.temporal = self.tpe(self.memory)
        return dict(bit_stream=stream.tobytes(), ec_parallel=ec)

    # ---- DMCLDProxy::decompress (dmc_ld_proxy.cpp:475-593)
    def decompress(self, bit_stream, qp, height, width, ec_parallel, reset_feature_memory):
        C = self.CH_Y
        zH, zW = (height + 63) // 64, (width + 63) // 64
        if self.memory_has_value:
            self.memory = self.fa_m(self.memory, self.feature_p)
        else:
            self.memory = self.fa_i(self.feature_i)
        self.temporal = self.tpe(self.memory)
        dec = orc_rans.Decoder(self.tables, np.frombuffer(bit_stream, dtype=np.uint8), ec_parallel)
        z_i8 = dec.decode_z(self.CH_Z * zH * zW, qp * self.CH_Z, self.CH_Z).reshape(zH, zW, self.CH_Z)
        common, q_dec, scales, means = self._priors(z_i8.astype(F16), qp)
        yH, yW = scales.shape[:2]
        idx, keep = sym.build_index_dec(scales, self.skip_thres)
        decoded = dec.decode_y(idx[keep])
        dec.close()
        self.ctx = self.fe(self.memory)
        y_q_r = sym.recover(decoded, keep, (yH, yW, C))
        mask_0, mask_1 = sym.get_mask_2x(yH, yW, C)
        y_hat = sym.restore_y(y_q_r, means, mask_0)
        means1 = self.spatial_prior(y_hat, common)
        y_hat = ((sym.restore_y(y_q_r, means1, mask_1) + y_hat).astype(F16) * sym.clamp_min_half(q_dec)).astype(F16)
        self.feature_p = self.decoder(y_hat, self.ctx, qp)
        head, x_hat = self.recon_head(self.feature_p)
        self.feature_i = head
        self.memory_has_value = not reset_feature_memory
        return x_hat


class DMCHTOracle:
    """CPU restatement of DMCHTSProxy / DMCHTLProxy (dmc_hts_proxy.cpp:492-710,
    dmc_htl_proxy.cpp:583-905): hierarchical inter codecs, 8 pictures per call. Networks:
    video_model_ht.py:26-317. HT-S codes all of y against one scale tensor in four masked steps
    (spatial prior predicts means only); HT-L runs the intra model's 4-step scheme (scales and
    means from the spatial prior, one symbol group per step)."""
    FRAMES, CH_SRC_I, CH_Y, CH_Z, CH_D, CH_M = 8, 192, 256, 128, 512, 512

    def __init__(self, state_dict, skip_thres, cdf_tables=None, is_hts=True):
        self.sd = to_np_state_dict(state_dict)
        self.net = Net(self.sd)
        self.is_hts = bool(is_hts)
        self.skip_thres = float(skip_thres)
        self.tables = orc_rans.Tables()
        if cdf_tables is not None:
            self.tables.set_cdf(cdf_tables[0], cdf_tables[1], 0)
            self.tables.set_cdf(cdf_tables[2], cdf_tables[3], 1)
        self.feature_i = self.memory = self.feature_p = self.ctx = None
        self.memory_has_value = False
        self.debug = {}

    # ---- sub-networks
    def fa_i(self, x):
        return self.net.chain(x, "feature_adaptor_i.conv.")

    def fa_m(self, memory, feature):
        return self.net.chain(np.concatenate([memory, feature], axis=-1), "feature_adaptor_m.conv.")

    def fe(self, memory):
        return self.net.chain(memory, "feature_extractor.conv.")

    def tpe(self, memory, qp):
        x = nn.mul_channel(memory, self.sd["q_feature"][qp])             # multiply_with_broadcast
        return self.net.rb_stride2(x, "temporal_prior_encoder.conv.", shortcut=not self.is_hts)

    def encoder(self, xu, ctx, qp):
        out = self.net.chain(np.concatenate([xu, ctx], axis=-1), "encoder.conv1.", q_last=self.sd["q_encoder"][qp])
        return nn.conv_kxk(out, self.sd["encoder.down.weight"], self.sd["encoder.down.bias"], 3, 2, 1)

    def hyper_encoder(self, y_pad):
        n, sc = self.net, not self.is_hts
        out = n.dcb(y_pad, "hyper_encoder.conv.0.")
        out = n.rb_stride2(out, "hyper_encoder.conv.1.", shortcut=sc)
        return n.rb_stride2(out, "hyper_encoder.conv.2.", shortcut=sc)

    def hyper_decoder(self, z_hat):
        n, sc = self.net, not self.is_hts
        out = n.rb_upsample(z_hat, "hyper_decoder.conv.0.", shortcut=sc)
        out = n.rb_upsample(out, "hyper_decoder.conv.1.", shortcut=sc)
        return n.dcb(out, "hyper_decoder.conv.2.")

    def prior_fusion(self, hyper, temporal):
        out = self.net.chain(np.concatenate([hyper, temporal], axis=-1), "y_prior_fusion.conv.")
        return self.net.conv1x1(out, "y_prior_fusion.conv.3.")

    def spatial_prior(self, y_hat_so_far, reduced, k):
        n = self.net
        out = n.dcb(np.concatenate([y_hat_so_far, reduced], axis=-1), "y_spatial_prior_adaptor_%d." % k)
        out = n.chain(out, "y_spatial_prior.conv.")
        return n.conv1x1(out, "y_spatial_prior.conv.3.")

    def decoder(self, y_hat, ctx, qp):
        up = self.net.subpel(y_hat, "decoder.up.")
        return self.net.chain(np.concatenate([up, ctx], axis=-1), "decoder.conv1.", q_last=self.sd["q_decoder"][qp])

    def recon_head(self, feature, only_last=False):
        """-> (list of 8 x_hat [H16*16, W16*16, 3], head output of picture 7 [H8, W8, 192])"""
        n = self.net
        outs, head = [], None
        common = None
        for i in range(self.FRAMES):
            if only_last and i != self.FRAMES - 1:
                continue
            if self.is_hts:
                if i % 2 == 0 or only_last:
                    common = n.dcb(feature, "recon_head.conv1.%d.0." % (i // 2))
                out = n.chain(common, "recon_head.conv2.%d." % i)
                head = n.conv1x1(out, "recon_head.conv2.%d.3." % i)
            else:
                out = n.chain(feature, "recon_head.conv.%d." % i)
                head = n.conv1x1(out, "recon_head.conv.%d.5." % i)
            outs.append(np.clip(nn.pixel_shuffle(head, 8), F16(-0.5), F16(0.5)).astype(F16))
        return outs, head

    # ---- add_ref_feature_from_frame (dmc_hts_proxy.cpp:492-502)
    def add_ref_feature_from_frame(self, frame, apply_adaptor):
        self.feature_i = nn.pixel_unshuffle(frame.astype(F16), 8)
        if apply_adaptor:
            self.memory = self.fa_i(self.feature_i)
            self.ctx = self.fe(self.memory)
        self.memory_has_value = bool(apply_adaptor)

    def _common(self, z_hat, qp):
        temporal = self.tpe(self.memory, qp)
        hyper = self.hyper_decoder(z_hat)[:temporal.shape[0], :temporal.shape[1]]
        common = self.prior_fusion(hyper, temporal)
        C = self.CH_Y
        return common, common[..., :C], common[..., C:2 * C], common[..., 2 * C:]

    # ---- compress (dmc_hts_proxy.cpp:504-585, dmc_htl_proxy.cpp:595-716)
    def compress(self, x, qp, reset_feature_memory):
        """x: fp16 [H, W, 24] (8 pictures x 3 planes, picture-major)."""
        H, W, _ = x.shape
        C = self.CH_Y
        pr, pb = get_padding_size(H, W, 16)
        xu = nn.pixel_unshuffle(replicate_pad(x.astype(F16), pb, pr), 8)
        y = self.encoder(xu, self.ctx, qp)
        yH, yW, _ = y.shape
        pr4, pb4 = get_padding_size(yH, yW, 4)
        z = self.hyper_encoder(replicate_pad(y, pb4, pr4))
        z_hat, z_i8 = sym.round_z(z)
        common, q_dec, scales, means = self._common(z_hat, qp)
        y = sym.divide_with_clamp(y, q_dec)
        reduced = self.net.conv1x1(common, "y_spatial_prior_reduction.")
        masks = sym.get_mask_4x(yH, yW, C)
        y_hat_so_far = None
        y_q_all = None
        y_syms = []
        for k in range(4):
            y_q, y_hat, s_hat = sym.process_with_mask(y, scales, means, masks[k], self.skip_thres)
            if not self.is_hts:
                comb, keep = sym.build_index_enc(sym.fold4(y_q), sym.fold4(s_hat), self.skip_thres)
                y_syms.append(comb[keep])
            y_q_all = y_q if y_q_all is None else (y_q_all + y_q).astype(F16)
            y_hat_so_far = y_hat if y_hat_so_far is None else (y_hat_so_far + y_hat).astype(F16)
            if k < 3:
                sp = self.spatial_prior(y_hat_so_far, reduced, k + 1)
                if self.is_hts:
                    means = sp
                else:
                    scales, means = sp[..., :C], sp[..., C:]
        y_hat = (y_hat_so_far * sym.clamp_min_half(q_dec)).astype(F16)
        if self.is_hts:
            comb, keep = sym.build_index_enc(y_q_all, scales, self.skip_thres)
            y_syms = [comb[keep]]
        total = sum(len(s) for s in y_syms)
        ec = compute_ec_parallel(total)
        segs = [("y", s) for s in reversed(y_syms)]
        segs.append(("z", z_i8.reshape(-1), qp * self.CH_Z, self.CH_Z))
        stream = orc_rans.encode(self.tables, segs, ec)
        self.debug = dict(y=y, y_hat=y_hat, z_i8=z_i8, y_syms=y_syms)
        # lambda_enc_1
        self.feature_p = self.decoder(y_hat, self.ctx, qp)
        if reset_feature_memory:
            _, head = self.recon_head(self.feature_p, only_last=True)     # forward_reset
            self.memory = self.fa_i(head)
        else:
            self.memory = self.fa_m(self.memory, self.feature_p)
        self.ctx = self.fe(self.memory)
        return dict(bit_stream=stream.tobytes(), ec_parallel=ec)

    # ---- decompress (dmc_hts_proxy.cpp:587-710, dmc_htl_proxy.cpp:718-905)
    def decompress(self, bit_stream, qp, height, width, ec_parallel, reset_feature_memory):
        C = self.CH_Y
        zH, zW = (height + 63) // 64, (width + 63) // 64
        if self.memory_has_value:
            self.memory = self.fa_m(self.memory, self.feature_p)
        else:
            self.memory = self.fa_i(self.feature_i)
        dec = orc_rans.Decoder(self.tables, np.frombuffer(bit_stream, dtype=np.uint8), ec_parallel)
        z_i8 = dec.decode_z(self.CH_Z * zH * zW, qp * self.CH_Z, self.CH_Z).reshape(zH, zW, self.CH_Z)
        common, q_dec, scales, means = self._common(z_i8.astype(F16), qp)
        yH, yW = scales.shape[:2]
        reduced = self.net.conv1x1(common, "y_spatial_prior_reduction.")
        self.ctx = self.fe(self.memory)
        masks = sym.get_mask_4x(yH, yW, C)
        y_hat_so_far = None
        if self.is_hts:
            idx, keep = sym.build_index_dec(scales, self.skip_thres)
            y_q_r = sym.recover(dec.decode_y(idx[keep]), keep, (yH, yW, C))
        for k in range(4):
            if self.is_hts:
                y_hat = sym.restore_y(y_q_r, means, masks[k])
            else:
                s_r = sym.fold4(np.where(masks[k], scales, F16(0)).astype(F16))
                idx, keep = sym.build_index_dec(s_r, self.skip_thres)
                y_q_g = sym.recover(dec.decode_y(idx[keep]), keep, (yH, yW, C // 4))
                y_hat = sym.restore_y_4x(y_q_g, means, masks[k])
            y_hat_so_far = y_hat if y_hat_so_far is None else (y_hat_so_far + y_hat).astype(F16)
            if k < 3:
                sp = self.spatial_prior(y_hat_so_far, reduced, k + 1)
                if self.is_hts:
                    means = sp
                else:
                    scales, means = sp[..., :C], sp[..., C:]
        dec.close()
        y_hat = (y_hat_so_far * sym.clamp_min_half(q_dec)).astype(F16)
        self.feature_p = self.decoder(y_hat, self.ctx, qp)
        x_hats, head = self.recon_head(self.feature_p)
        self.feature_i = head
        self.memory_has_value = not reset_feature_memory
        return x_hats
