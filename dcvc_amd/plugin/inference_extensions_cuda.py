"""Drop-in for the reference's native plugin ``inference_extensions_cuda``
(/root/reference/src/layers/extensions/inference/bind.cpp:11-39) on MI355X: the same class and
method names with the same argument meaning, implemented over the C ABI of libdcvc_amd.so
(include/dcvc_amd_codec.h). The reference imports it lazily by this name
(src/models/image_model.py:197); put ``dcvc_amd/plugin`` on ``sys.path`` or call
``dcvc_amd.install_plugin()``.

There is no fallback of any kind: a missing shared object or a failing call raises.
"""
import ctypes
import os
import warnings

import numpy as np
import torch


def _hw_queue_policy():
    """ONE hardware queue per stream-priority level (ROCclr's GPU_MAX_HW_QUEUES; default 4) is what the codec objects were
    measured with: every object brings a compute stream and a transfer stream, and with the runtime's default the throughput
    of several objects in one process depended on the ORDER they were created in - up to 2x (profiles/r05_hw_queues.txt). The
    variable is read when the HIP runtime starts, so:
      * it is "1" already                       -> nothing to do ("set by the host")
      * unset and the runtime has not started   -> set here ("set by the plug-in"); not with several ranks in the job (RCCL's
                                                   stream would share the queue with the compute streams: never measured)
      * anything else                           -> the runtime is up with its default (the reference harness imports this module
                                                   lazily, behind its first CUDA call) or the host chose another value: WARN once,
                                                   the drop-in must not silently run at half speed
    Returns the policy string (also `hw_queue_policy` of this module; bench.py reports it)."""
    val = os.environ.get("GPU_MAX_HW_QUEUES")
    if val == "1":
        return "GPU_MAX_HW_QUEUES=1 (set by the host)"
    several_ranks = int(os.environ.get("WORLD_SIZE", "1") or "1") > 1
    if val is None and not several_ranks and not torch.cuda.is_initialized():
        os.environ["GPU_MAX_HW_QUEUES"] = "1"
        return "GPU_MAX_HW_QUEUES=1 (set by the plug-in before the HIP runtime started)"
    if several_ranks:
        return "GPU_MAX_HW_QUEUES=%s (several ranks: left alone)" % val
    warnings.warn("dcvc_amd: the HIP runtime %s; the codec objects were measured with GPU_MAX_HW_QUEUES=1 and can run up to 2x "
                  "slower otherwise (INTEGRATION.md, section 1). Export GPU_MAX_HW_QUEUES=1 before the first CUDA / HIP call of "
                  "the process." % ("is already up with its default of 4 hardware queues per priority level" if val is None
                                    else "was given GPU_MAX_HW_QUEUES=%s" % val), RuntimeWarning, stacklevel=3)
    return "GPU_MAX_HW_QUEUES=%s (NOT the measured setting)" % val


hw_queue_policy = _hw_queue_policy()

from dcvc_amd import _lib  # noqa: E402  (behind the policy: loading the library may start the runtime)

_vp, _ci = ctypes.c_void_p, ctypes.c_int
_F = dict(
    create=_lib.fn("dcvc_dmci_create", _vp, []),
    destroy=_lib.fn("dcvc_dmci_destroy", None, [_vp]),
    set_param=_lib.fn("dcvc_dmci_set_param", _ci,
                      [_vp, _ci, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_vp),
                       ctypes.POINTER(_ci), ctypes.POINTER(_ci), ctypes.POINTER(ctypes.c_int64),
                       ctypes.c_float]),
    compress=_lib.fn("dcvc_dmci_compress", _ci, [_vp, _vp, _ci, _ci, _ci, _ci, _ci, _vp, _vp]),
    get_stream=_lib.fn("dcvc_dmci_get_stream", ctypes.c_int64, [_vp, _vp, ctypes.c_size_t]),
    decompress=_lib.fn("dcvc_dmci_decompress", _ci, [_vp, _vp, ctypes.c_size_t, _ci, _ci, _ci, _ci, _vp, _vp]),
    use_graphs=_lib.fn("dcvc_dmci_set_use_graphs", _ci, [_vp, _ci]),
    debug_read=_lib.fn("dcvc_dmci_debug_read", ctypes.c_int64, [_vp, ctypes.c_char_p, _vp, ctypes.c_size_t, _vp]),
)

_SET_PARAM_ARGS = [_vp, _ci, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_vp), ctypes.POINTER(_ci),
                   ctypes.POINTER(_ci), ctypes.POINTER(ctypes.c_int64), ctypes.c_float]
_LD = dict(
    create=_lib.fn("dcvc_dmcld_create", _vp, []),
    destroy=_lib.fn("dcvc_dmcld_destroy", None, [_vp]),
    set_param=_lib.fn("dcvc_dmcld_set_param", _ci, _SET_PARAM_ARGS),
    add_ref=_lib.fn("dcvc_dmcld_add_ref_feature_from_frame", _ci, [_vp, _vp, _ci, _ci, _ci, _vp]),
    compress=_lib.fn("dcvc_dmcld_compress", _ci, [_vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp]),
    get_stream=_lib.fn("dcvc_dmcld_get_stream", ctypes.c_int64, [_vp, _vp, ctypes.c_size_t]),
    decompress=_lib.fn("dcvc_dmcld_decompress", _ci,
                       [_vp, _vp, ctypes.c_size_t, _ci, _ci, _ci, _ci, _ci, _vp, _vp]),
    use_graphs=_lib.fn("dcvc_dmcld_set_use_graphs", _ci, [_vp, _ci]),
    export_state=_lib.fn("dcvc_dmcld_export_state", ctypes.c_int64, [_vp, _vp, ctypes.c_size_t, _vp]),
    import_state=_lib.fn("dcvc_dmcld_import_state", _ci, [_vp, _vp, ctypes.c_size_t, _ci, _ci, _vp]),
    debug_read=_lib.fn("dcvc_dmcld_debug_read", ctypes.c_int64, [_vp, ctypes.c_char_p, _vp, ctypes.c_size_t, _vp]),
)

_HT = dict(
    create=_lib.fn("dcvc_dmcht_create", _vp, [_ci]),
    destroy=_lib.fn("dcvc_dmcht_destroy", None, [_vp]),
    set_param=_lib.fn("dcvc_dmcht_set_param", _ci, _SET_PARAM_ARGS),
    add_ref=_lib.fn("dcvc_dmcht_add_ref_feature_from_frame", _ci, [_vp, _vp, _ci, _ci, _ci, _vp]),
    compress=_lib.fn("dcvc_dmcht_compress", _ci, [_vp, _vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp]),
    get_stream=_lib.fn("dcvc_dmcht_get_stream", ctypes.c_int64, [_vp, _vp, ctypes.c_size_t]),
    decompress=_lib.fn("dcvc_dmcht_decompress", _ci,
                       [_vp, _vp, ctypes.c_size_t, _ci, _ci, _ci, _ci, _ci, _vp, _vp]),
    use_graphs=_lib.fn("dcvc_dmcht_set_use_graphs", _ci, [_vp, _ci]),
    export_state=_lib.fn("dcvc_dmcht_export_state", ctypes.c_int64, [_vp, _vp, ctypes.c_size_t, _vp]),
    import_state=_lib.fn("dcvc_dmcht_import_state", _ci, [_vp, _vp, ctypes.c_size_t, _ci, _ci, _vp]),
    set_recon_mask=_lib.fn("dcvc_dmcht_set_recon_mask", _ci, [_vp, ctypes.c_uint]),
    export_feature=_lib.fn("dcvc_dmcht_export_feature", ctypes.c_int64, [_vp, _vp, ctypes.c_size_t, _vp]),
    import_feature=_lib.fn("dcvc_dmcht_import_feature", _ci, [_vp, _vp, ctypes.c_size_t, _ci, _ci, _vp]),
    run_recon_heads=_lib.fn("dcvc_dmcht_run_recon_heads", _ci, [_vp, ctypes.c_uint, _vp, _vp]),
    debug_read=_lib.fn("dcvc_dmcht_debug_read", ctypes.c_int64, [_vp, ctypes.c_char_p, _vp, ctypes.c_size_t, _vp]),
)

_DTYPES = {torch.float16: 0, torch.float32: 1, torch.int32: 2}


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pack_state_dict(state_dict):
    """state_dict (torch tensors on any device) -> ctypes arrays over host copies."""
    names, keep, ptrs, dtypes, ndims, dims = [], [], [], [], [], []
    for name, t in state_dict.items():
        if not torch.is_tensor(t):
            t = torch.as_tensor(t)
        if t.dtype == torch.int64:
            t = t.to(torch.int32)
        if t.dtype not in _DTYPES:
            t = t.float()
        h = t.detach().to("cpu").contiguous()
        keep.append(h)
        names.append(name.encode())
        ptrs.append(h.data_ptr())
        dtypes.append(_DTYPES[h.dtype])
        ndims.append(h.dim())
        dims.extend(h.shape)
    n = len(names)
    return (n, (ctypes.c_char_p * n)(*names), (_vp * n)(*ptrs), (_ci * n)(*dtypes), (_ci * n)(*ndims),
            (ctypes.c_int64 * max(1, len(dims)))(*dims), keep)


def _nhwc_ptr(x, channels):
    """[1, C, H, W] channels_last fp16 CUDA tensor -> pointer to its [H][W][C] memory."""
    if x.dim() != 4 or x.shape[0] != 1 or x.shape[1] != channels:
        raise ValueError("expected a [1, %d, H, W] tensor, got %s" % (channels, tuple(x.shape)))
    if x.dtype != torch.float16 or not x.is_cuda:
        raise ValueError("expected a CUDA fp16 tensor")
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    return x, ctypes.c_void_p(x.data_ptr())


class _Proxy:
    """Handle ownership + the calls every proxy class has."""
    _FN = None

    _CREATE_ARGS = ()

    def __init__(self):
        self._h = self._FN["create"](*self._CREATE_ARGS)
        if not self._h:
            raise _lib.DcvcError(_lib.lib().dcvc_last_error().decode())
        self._x_hat = None
        self._destroy = self._FN["destroy"]          # survives module teardown at interpreter exit

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._destroy(h)

    def set_param(self, state_dict, skip_thres):
        n, names, ptrs, dtypes, ndims, dims, keep = _pack_state_dict(state_dict)
        _lib.check(self._FN["set_param"](self._h, n, names, ptrs, dtypes, ndims, dims, float(skip_thres)))
        del keep

    def _stream_bytes(self):
        n = self._FN["get_stream"](self._h, None, 0)
        out = np.empty(n, dtype=np.uint8)
        self._FN["get_stream"](self._h, out.ctypes.data_as(_vp), n)
        return out

    # ---- not part of the reference surface
    def set_use_graphs(self, on):
        _lib.check(self._FN["use_graphs"](self._h, 1 if on else 0))

    def debug_read(self, name, dtype):
        n = _lib.check(self._FN["debug_read"](self._h, name.encode(), None, 0, _stream_ptr()))
        buf = np.empty(n, dtype=np.uint8)
        _lib.check(self._FN["debug_read"](self._h, name.encode(), buf.ctypes.data_as(_vp), n, _stream_ptr()))
        return buf.view(dtype)

    def _out_buffer(self, height, width, device):
        h16, w16 = (height + 15) // 16 * 16, (width + 15) // 16 * 16
        if self._x_hat is None or tuple(self._x_hat.shape) != (1, 3, h16, w16) or self._x_hat.device != device:
            # proxy-owned, overwritten by the next call (dmci_proxy.cpp m_x_hat)
            self._x_hat = torch.empty((1, 3, h16, w16), dtype=torch.float16, device=device).contiguous(
                memory_format=torch.channels_last)
        return self._x_hat



class DMCIProxy(_Proxy):
    """bind.cpp:12-16 / dmci_proxy.h:134-150."""
    _FN = _F

    def compress(self, x, qp, padding_b, padding_r):
        """-> (np.ndarray[uint8] bit stream, x_hat [1, 3, ceil16(H), ceil16(W)], ec_parallel)"""
        x, xp = _nhwc_ptr(x, 3)
        height, width = int(x.shape[2]), int(x.shape[3])
        x_hat = self._out_buffer(height, width, x.device)
        ec = _lib.check(_F["compress"](self._h, xp, height, width, int(qp), int(padding_b), int(padding_r),
                                       ctypes.c_void_p(x_hat.data_ptr()), _stream_ptr()))
        return self._stream_bytes(), x_hat, int(ec)

    def decompress(self, bit_stream, qp, height, width, entropy_coder_parallel):
        bs = np.ascontiguousarray(bit_stream, dtype=np.uint8)
        device = torch.device("cuda", torch.cuda.current_device())
        x_hat = self._out_buffer(int(height), int(width), device)
        _lib.check(_F["decompress"](self._h, bs.ctypes.data_as(_vp), bs.size, int(qp), int(height), int(width),
                                    int(entropy_coder_parallel), ctypes.c_void_p(x_hat.data_ptr()),
                                    _stream_ptr()))
        return x_hat


class DMCLDProxy(_Proxy):
    """bind.cpp:31-38 / dmc_ld_proxy.h: the low-delay inter codec."""
    _FN = _LD

    def add_ref_feature_from_frame(self, frame, apply_feature_adaptor=True):
        frame, fp = _nhwc_ptr(frame, 3)
        _lib.check(_LD["add_ref"](self._h, fp, int(frame.shape[2]), int(frame.shape[3]),
                                  1 if apply_feature_adaptor else 0, _stream_ptr()))

    def compress(self, x, qp, reset_feature_memory, padding_b, padding_r):
        """-> (np.ndarray[uint8] bit stream, ec_parallel)"""
        x, xp = _nhwc_ptr(x, 3)
        ec = _lib.check(_LD["compress"](self._h, xp, int(x.shape[2]), int(x.shape[3]), int(qp),
                                        1 if reset_feature_memory else 0, int(padding_b), int(padding_r),
                                        _stream_ptr()))
        return self._stream_bytes(), int(ec)

    def decompress(self, bit_stream, qp, height, width, entropy_coder_parallel, reset_feature_memory):
        bs = np.ascontiguousarray(bit_stream, dtype=np.uint8)
        device = torch.device("cuda", torch.cuda.current_device())
        x_hat = self._out_buffer(int(height), int(width), device)
        _lib.check(_LD["decompress"](self._h, bs.ctypes.data_as(_vp), bs.size, int(qp), int(height),
                                     int(width), int(entropy_coder_parallel),
                                     1 if reset_feature_memory else 0,
                                     ctypes.c_void_p(x_hat.data_ptr()), _stream_ptr()))
        return x_hat


def _export_state(self):
    """-> uint8 CUDA tensor holding the temporal state (send it with torch.distributed.send)."""
    n = _lib.check(self._FN["export_state"](self._h, None, 0, _stream_ptr()))
    buf = torch.empty(n, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
    _lib.check(self._FN["export_state"](self._h, ctypes.c_void_p(buf.data_ptr()), n, _stream_ptr()))
    return buf


def _import_state(self, state, height, width):
    """state: uint8 CUDA tensor from export_state() of a codec with the same parameters."""
    if state.dtype != torch.uint8 or not state.is_cuda or not state.is_contiguous():
        raise ValueError("expected a contiguous uint8 CUDA tensor")
    _lib.check(self._FN["import_state"](self._h, ctypes.c_void_p(state.data_ptr()), state.numel(), int(height),
                                        int(width), _stream_ptr()))


DMCLDProxy.export_state = _export_state
DMCLDProxy.import_state = _import_state


class _DMCHTProxy(_Proxy):
    """The hierarchical inter codecs: 8 pictures per call (video_model_ht.py:16 g_frame_delay)."""
    _FN = _HT
    FRAMES = 8

    def add_ref_feature_from_frame(self, frame, apply_feature_adaptor=True):
        frame, fp = _nhwc_ptr(frame, 3)
        _lib.check(_HT["add_ref"](self._h, fp, int(frame.shape[2]), int(frame.shape[3]),
                                  1 if apply_feature_adaptor else 0, _stream_ptr()))

    def compress(self, x, qp, reset_feature_memory, padding_b, padding_r):
        """x: [1, 24, H, W] (8 pictures x 3 planes) -> (np.ndarray[uint8] bit stream, ec_parallel)"""
        x, xp = _nhwc_ptr(x, 3 * self.FRAMES)
        ec = _lib.check(_HT["compress"](self._h, xp, int(x.shape[2]), int(x.shape[3]), int(qp),
                                        1 if reset_feature_memory else 0, int(padding_b), int(padding_r),
                                        _stream_ptr()))
        return self._stream_bytes(), int(ec)

    def decompress(self, bit_stream, qp, height, width, entropy_coder_parallel, reset_feature_memory):
        """-> list of 8 x_hat [1, 3, ceil16(H), ceil16(W)] (proxy-owned, overwritten by the next call)"""
        bs = np.ascontiguousarray(bit_stream, dtype=np.uint8)
        device = torch.device("cuda", torch.cuda.current_device())
        h16, w16 = (int(height) + 15) // 16 * 16, (int(width) + 15) // 16 * 16
        if self._x_hat is None or tuple(self._x_hat.shape) != (self.FRAMES, 3, h16, w16) or self._x_hat.device != device:
            self._x_hat = torch.empty((self.FRAMES, 3, h16, w16), dtype=torch.float16, device=device).contiguous(
                memory_format=torch.channels_last)
        _lib.check(_HT["decompress"](self._h, bs.ctypes.data_as(_vp), bs.size, int(qp), int(height),
                                     int(width), int(entropy_coder_parallel),
                                     1 if reset_feature_memory else 0,
                                     ctypes.c_void_p(self._x_hat.data_ptr()), _stream_ptr()))
        return self._pictures(self._recon_mask)

    _recon_mask = 0xFF

    def _pictures(self, mask):
        """the 8 entries of a decompress() / run_recon_heads() result: pictures whose head did not run are None, not a
        view of a buffer nobody wrote (advisor, round 3: a plain decompress() behind a fan-out returned stale tensors)"""
        return [self._x_hat[i:i + 1] if (mask >> i) & 1 else None for i in range(self.FRAMES)]

    # ---- reconstruction-head fan-out over several GPUs (not part of the reference surface; SURVEY 8e iii)
    def set_recon_mask(self, mask):
        """decompress() runs only the heads of the pictures in bit mask `mask` (the other entries of the
        returned list are then not written)."""
        _lib.check(_HT["set_recon_mask"](self._h, int(mask) & 0xFF))
        self._recon_mask = int(mask) & 0xFF

    def recon_mask(self):
        return self._recon_mask

    def export_feature(self):
        """feature_p of the last decompress() as a dense fp16 device tensor [P8, 512]"""
        n = _lib.check(_HT["export_feature"](self._h, None, 0, _stream_ptr()))
        t = torch.empty(int(n) // 2, dtype=torch.float16, device=torch.device("cuda", torch.cuda.current_device()))
        _lib.check(_HT["export_feature"](self._h, ctypes.c_void_p(t.data_ptr()), int(n), _stream_ptr()))
        return t

    def import_feature(self, feature, height, width):
        if not feature.is_cuda or not feature.is_contiguous() or feature.dtype != torch.float16:
            raise ValueError("expected a contiguous fp16 CUDA tensor")
        _lib.check(_HT["import_feature"](self._h, ctypes.c_void_p(feature.data_ptr()), feature.numel() * 2, int(height),
                                         int(width), _stream_ptr()))

    def run_recon_heads(self, mask, height, width):
        """heads of the pictures in `mask` on the imported / own feature_p -> list of 8 x_hat (only those of
        the mask are written)"""
        device = torch.device("cuda", torch.cuda.current_device())
        h16, w16 = (int(height) + 15) // 16 * 16, (int(width) + 15) // 16 * 16
        if self._x_hat is None or tuple(self._x_hat.shape) != (self.FRAMES, 3, h16, w16) or self._x_hat.device != device:
            self._x_hat = torch.empty((self.FRAMES, 3, h16, w16), dtype=torch.float16, device=device).contiguous(
                memory_format=torch.channels_last)
        _lib.check(_HT["run_recon_heads"](self._h, int(mask) & 0xFF, ctypes.c_void_p(self._x_hat.data_ptr()), _stream_ptr()))
        return self._pictures(int(mask) & 0xFF)


_DMCHTProxy.export_state = _export_state
_DMCHTProxy.import_state = _import_state


class DMCHTSProxy(_DMCHTProxy):
    """bind.cpp:17-23 / dmc_hts_proxy.h."""
    _CREATE_ARGS = (1,)


class DMCHTLProxy(_DMCHTProxy):
    """bind.cpp:24-30 / dmc_htl_proxy.h."""
    _CREATE_ARGS = (0,)
