"""Helpers for the -m gpu tests: typed ctypes access to the kernel-level C ABI
(include/dcvc_amd_ops.h) with torch tensors as device memory."""
import ctypes

import torch

from dcvc_amd import _lib

vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def _f(name, args):
    return _lib.fn(name, ci, args)


class Ops:
    def __init__(self):
        self.conv1x1 = _f("dcvc_conv1x1", [vp, ci, vp, vp, vp, ci, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp])
        self.conv_kxk = _f("dcvc_conv_kxk", [vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp])
        self.tconv2x2 = _f("dcvc_tconv2x2", [vp, ci, vp, vp, ci, ci, ci, ci, ci, vp])
        self.dwconv3x3 = _f("dcvc_dwconv3x3", [vp, ci, vp, vp, ci, ci, ci, ci, vp])
        self.pad_unshuffle8 = _f("dcvc_pad_unshuffle8", [vp, ci, ci, ci, vp, ci, ci, vp])
        self.shuffle8 = _f("dcvc_shuffle8", [vp, ci, ci, ci, ci, ci, vp, vp])
        self.shuffle2 = _f("dcvc_shuffle2", [vp, ci, ci, ci, ci, vp, ci, vp])
        self.replicate_pad = _f("dcvc_replicate_pad", [vp, ci, ci, ci, ci, ci, ci, vp, ci, vp])
        self.crop = _f("dcvc_crop", [vp, ci, ci, vp, ci, ci, ci, ci, vp])
        self.mul_channel = _f("dcvc_mul_channel", [vp, ci, vp, vp, ci, ci, ci, vp])
        self.round_z = _f("dcvc_round_z", [vp, vp, vp, ci, vp])
        self.int8_to_half = _f("dcvc_int8_to_half", [vp, vp, ci, vp])
        self.symbol_blocks = _f("dcvc_symbol_blocks", [ci])
        self.y_step_enc = _f("dcvc_y_step_enc", [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp,
                                                 ci, ci, ci, ci, cf, vp])
        self.y_step_dec_index = _f("dcvc_y_step_dec_index", [vp, ci, vp, vp, vp, vp, vp,
                                                             ci, ci, ci, ci, cf, vp])
        self.y_step_dec_restore = _f("dcvc_y_step_dec_restore", [vp, vp, vp, vp, vp, ci, vp, ci,
                                                                 ci, ci, ci, ci, vp])
        self.mask_step_enc = _f("dcvc_mask_step_enc", [vp, ci, vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp,
                                                       ci, ci, ci, ci, ci, cf, vp])
        self.mask_dec_index = _f("dcvc_mask_dec_index", [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, cf, vp])
        self.mask_step_dec = _f("dcvc_mask_step_dec", [vp, vp, vp, vp, vp, vp, ci, vp, ci, vp, ci,
                                                       ci, ci, ci, ci, ci, vp])
        self.ffn_fused = _f("dcvc_ffn_fused", [vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, vp])
        self.dcb_tail = _f("dcvc_dcb_tail", [vp, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci,
                                             ci, ci, ci, ci, ci, ci, vp])
        self.dcb_nsplit = _f("dcvc_dcb_nsplit", [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, ci,
                                                 ci, ci, ci, ci, vp])
        self.dcb_nsplit_fin = _f("dcvc_dcb_nsplit_fin", [vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci,
                                                         vp, ci, ci, ci, ci, ci, vp])
        self.dcb_nsplit_fin_supported = _f("dcvc_dcb_nsplit_fin_supported", [ci, ci, ci])
        self.dcb_nsplit_dw = _f("dcvc_dcb_nsplit_dw", [vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci,
                                                       vp, vp, vp, vp, ci, ci, vp, ci, ci, ci, ci, ci, vp])
        self.dcb_nsplit_dw_supported = _f("dcvc_dcb_nsplit_dw_supported", [ci, ci, ci])
        self.dcb_pair = _f("dcvc_dcb_pair", [vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci, vp])
        self.dcb_pair_supported = _f("dcvc_dcb_pair_supported", [ci, ci, ci])
        self.dcb_nsplit_pack = _f("dcvc_dcb_nsplit_pack", [vp, vp, vp, vp, ci, ci, vp, ctypes.POINTER(vp)])
        self.dcb_nsplit_packed = _f("dcvc_dcb_nsplit_packed", [vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp, ci,
                                                               ci, ci, ci, vp])
        self.dcb_nsplit_free = _f("dcvc_dcb_nsplit_free", [vp])
        self.scale_clamped = _f("dcvc_scale_clamped", [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp])


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(fn, *args):
    _lib.check(fn(*args))


def nhwc(t):
    """[1, C, H, W] tensor -> contiguous [H, W, C]."""
    return t[0].permute(1, 2, 0).contiguous()


def nchw(t):
    """[H, W, C] -> [1, C, H, W]."""
    return t.permute(2, 0, 1).unsqueeze(0).contiguous()
