"""Host-side mirror of the reference's Python operator surface for the inference hot path
(/root/reference/src/models/image_model.py ``DMCI``, common_model.py ``CompressionModel``,
entropy_models.py): same class / method names, argument meaning and state_dict keys, so the
reference's harness logic (test_video.py:167-399) drops onto it. The networks themselves run in
libdcvc_amd.so; this module only holds parameters, builds the entropy-coder CDF tables and
forwards ``compress`` / ``decompress`` to the proxy classes of ``inference_extensions_cuda``.

There is no training graph here (``forward_one_frame`` is out of scope) and no CPU
fallback: without the built extension ``compress`` raises NotImplementedError exactly like the
reference does (image_model.py:196-202).
"""
import math

import numpy as np
import torch
from torch import nn

import dcvc_amd
from dcvc_amd import arch

MAX_ENTROPY_CODING_VALUE = 8          # entropy_models.py:12


class _Node(nn.Module):
    """Anonymous container so that flat names like 'enc.enc_2.0.dc.3.weight' become real nested
    parameters (state_dict keys identical to the reference modules')."""


def _register(root, name, tensor):
    parts = name.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
    """entropy_models.py:45-75: reorder to 0,+1,-1,+2,... then quantise each row."""
    from MLCodec_extensions_cpp import pmf_to_quantized_cdf
    cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
    for i in range(len(pmf_length)):
        n = int(pmf_length[i])
        prob = torch.cat((pmf[i][:n], tail_mass[i]), dim=0)
        center = (prob.numel() - 1) // 2
        order = [center]
        for k in range(1, center + 1):
            order += [center + k, center - k]
        order.append(prob.numel() - 1)                        # the tail (escape) mass stays last
        row = pmf_to_quantized_cdf(prob[order].tolist())
        cdf[i, :len(row)] = torch.tensor(row, dtype=torch.int32)
    return cdf


def _bit_estimator_cdf(h, b, a):
    """Factorised prior of z: restates BitEstimator.update (entropy_models.py:113-149) incl.
    bit_estimator_z_prob (layers.py:13-19)."""
    import torch.nn.functional as F
    # on the parameters' own device, like the reference (update() runs on whatever device the
    # model lives on): softplus / tanh / sigmoid may differ in the last place between devices,
    # which can move a 0.001 / 0.999 range cut or a quantised frequency
    dev = h.device
    h, b, a = h.float(), b.float(), a.float()
    qp_num, channel, _ = h.shape

    def prob(x):
        for i in range(4):
            x = x * F.softplus(h[:, :, i:i + 1, None]) + b[:, :, i:i + 1, None]
            if i != 3:
                x = x + torch.tanh(x) * torch.tanh(a[:, :, i:i + 1, None])
        return torch.sigmoid(x)

    zeros = torch.zeros((qp_num, channel, 1, 1), device=dev)
    sym_range = zeros + MAX_ENTROPY_CODING_VALUE
    for i in range(MAX_ENTROPY_CODING_VALUE, 1, -1):
        neg, pos = prob(zeros - i), prob(zeros + i)
        sym_range = torch.where(torch.logical_and(neg < 0.001, pos > 0.999),
                                torch.tensor(float(i), device=dev), sym_range)
    sym_range = sym_range.int()
    pmf_length = sym_range * 2 + 1
    max_length = MAX_ENTROPY_CODING_VALUE * 2 + 1
    samples = torch.arange(max_length, device=dev)[None, None, None, :] - sym_range
    lower, upper = prob(samples - 0.5), prob(samples + 0.5)
    pmf = (upper - lower)[:, :, 0, :]
    upper_r = prob(sym_range.float())
    tail_mass = lower[:, :, 0, :1] + (1.0 - upper_r[:, :, 0, -1:])
    pmf = pmf.reshape([-1, max_length]).cpu()
    tail_mass = tail_mass.reshape([-1, 1]).cpu()
    pmf_length = pmf_length.reshape([-1]).cpu()
    cdf = _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
    return cdf.numpy(), (pmf_length + 2).int().numpy()


def _gaussian_cdf():
    """128-level Gaussian scale table 0.11 .. 16: restates GaussianEncoder.update
    (entropy_models.py:152-217)."""
    scale_table = torch.exp(torch.linspace(math.log(0.11), math.log(16.0), 128))
    zeros = torch.zeros_like(scale_table)
    sym_range = zeros + MAX_ENTROPY_CODING_VALUE
    dist = torch.distributions.normal.Normal(0., scale_table)
    for i in range(MAX_ENTROPY_CODING_VALUE, 1, -1):
        probs = torch.squeeze(dist.cdf(zeros + i))
        sym_range = torch.where(probs > 0.999, torch.tensor(float(i)), sym_range)
    sym_range = sym_range.int()
    pmf_length = 2 * sym_range + 1
    max_length = 2 * MAX_ENTROPY_CODING_VALUE + 1
    samples = (torch.arange(max_length) - sym_range[:, None]).float()
    dist = torch.distributions.normal.Normal(0., scale_table[:, None])
    upper, lower = dist.cdf(samples + 0.5), dist.cdf(samples - 0.5)
    pmf = upper - lower
    tail_mass = 2 * lower[:, :1]
    cdf = _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
    return cdf.numpy(), (pmf_length + 2).int().numpy()


class CompressionModel(nn.Module):
    """common_model.py:33-210, inference subset."""
    _SPEC = None
    _PROXY = None

    def __init__(self):
        super().__init__()
        dcvc_amd.install_plugin()
        for name, shape in self._spec().items():
            _register(self, name, torch.zeros(shape))
        self.proxy = None
        self.skip_thres = 0.0
        self._cdf = None

    def _spec(self):
        return self._SPEC()

    @staticmethod
    def qp_num():
        return arch.QP_NUM

    @staticmethod
    def get_padding_size(height, width, p=64):
        new_h = (height + p - 1) // p * p
        new_w = (width + p - 1) // p * p
        return new_w - width, new_h - height          # (padding_right, padding_bottom)

    def update(self, skip_thres):
        """Builds the quantised CDF tables of both entropy models (common_model.py:152-155)."""
        self.skip_thres = float(skip_thres)
        be = self.bit_estimator_z
        z_cdf, z_len = _bit_estimator_cdf(be.h.data, be.b.data, be.a.data)
        y_cdf, y_len = _gaussian_cdf()
        self._cdf = {
            "gaussian_encoder.quantized_cdf": y_cdf, "gaussian_encoder.cdf_length": y_len,
            "bit_estimator_z.quantized_cdf": z_cdf, "bit_estimator_z.cdf_length": z_len,
        }
        self.proxy = None

    def set_cdf_info(self, z_cdf, z_len, y_cdf, y_len):
        """Installs precomputed tables instead of update()'s. Encoder and decoder must use the SAME
        tables, and update() evaluates softplus / tanh / sigmoid / erf on whatever device and CPU it
        runs on: measured here, an Intel and an AMD host build tables that differ in a few entries
        from identical parameters (the reference has the same property, it builds them on the GPU,
        entropy_models.py:113-217). Ship the tables with the stream's model when hosts differ."""
        i32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.int32)
        self._cdf = {
            "gaussian_encoder.quantized_cdf": i32(y_cdf), "gaussian_encoder.cdf_length": i32(y_len),
            "bit_estimator_z.quantized_cdf": i32(z_cdf), "bit_estimator_z.cdf_length": i32(z_len),
        }
        self.proxy = None

    def get_cdf_info(self):
        c = self._cdf
        return (c["bit_estimator_z.quantized_cdf"], c["bit_estimator_z.cdf_length"],
                c["gaussian_encoder.quantized_cdf"], c["gaussian_encoder.cdf_length"])

    def add_cdf_to_state_dict(self, state_dict):
        """common_model.py:64-70."""
        if self._cdf is None:
            raise RuntimeError("call update(skip_thres) before compress()/decompress()")
        state_dict.update({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in self._cdf.items()})
        return state_dict

    def load_state_dict(self, state_dict, *args, **kwargs):
        res = super().load_state_dict(state_dict, *args, **kwargs)
        self.proxy = None                      # cached native parameters are stale now
        return res

    def _ensure_proxy(self):
        if self.proxy is None:
            try:
                import inference_extensions_cuda as ext
            except Exception as e:       # same contract as image_model.py:196-202
                raise NotImplementedError(
                    "cannot import the MI355X implementation for inference "
                    "(build it with `python -m dcvc_amd.build`): %s" % e)
            state_dict = self.add_cdf_to_state_dict(self.state_dict())
            self.proxy = getattr(ext, self._PROXY)()
            self.proxy.set_param(state_dict, self.skip_thres)
        return self.proxy


class DMCI(CompressionModel):
    """image_model.py:126-217 (inference subset)."""
    _SPEC = staticmethod(arch.dmci_spec)
    _PROXY = "DMCIProxy"

    def compress(self, x, qp, padding_b, padding_r):
        bit_stream, x_hat, ec_parallel = self._ensure_proxy().compress(x, qp, padding_b, padding_r)
        return {"bit_stream": bit_stream.tobytes(), "x_hat": x_hat, "ec_parallel": ec_parallel}

    def decompress(self, bit_stream, sps, qp, ec_part):
        x_hat = self._ensure_proxy().decompress(
            np.frombuffer(bit_stream, dtype=np.uint8), qp, sps["height"], sps["width"], ec_part)
        return {"x_hat": x_hat}


class DMC(CompressionModel):
    """video_model_ld.py:191-308 (inference subset): the low-delay inter model. The temporal
    state (reference feature, memory, context) lives inside the native proxy."""
    _SPEC = staticmethod(arch.dmc_ld_spec)
    _PROXY = "DMCLDProxy"

    def clear_dpb(self):
        """video_model_ld.py:226-229; the native state is overwritten by the next
        add_ref_feature_from_frame, so there is nothing to release."""

    def add_ref_feature_from_frame(self, frame, apply_feature_adaptor=True):
        return self._ensure_proxy().add_ref_feature_from_frame(frame, apply_feature_adaptor)

    def compress(self, x, qp, reset_feature_memory, padding_b, padding_r):
        bit_stream, ec_parallel = self._ensure_proxy().compress(
            x, qp, bool(reset_feature_memory), padding_b, padding_r)
        return {"bit_stream": bit_stream.tobytes(), "ec_parallel": ec_parallel}

    def decompress(self, bit_stream, sps, qp, ec_part, reset_feature_memory):
        x_hat = self._ensure_proxy().decompress(
            np.frombuffer(bit_stream, dtype=np.uint8), qp, sps["height"], sps["width"], ec_part,
            bool(reset_feature_memory))
        return {"x_hat": x_hat}


class DMCHT(DMC):
    """video_model_ht.py:320-470 (inference subset): the hierarchical inter models, 8 pictures per
    call. ``model_structure`` is "hts" (DMCHTSProxy) or "htl" (DMCHTLProxy); the reference spells
    it ``DMC(ModelStructure.HTS)`` in its own module."""

    def __init__(self, model_structure="hts"):
        ms = str(getattr(model_structure, "name", model_structure)).lower()
        if ms not in ("hts", "htl"):
            raise ValueError("model_structure must be 'hts' or 'htl'")
        self.is_hts = ms == "hts"
        self._PROXY = "DMCHTSProxy" if self.is_hts else "DMCHTLProxy"
        super().__init__()

    def _spec(self):
        return arch.dmc_ht_spec(self.is_hts)
