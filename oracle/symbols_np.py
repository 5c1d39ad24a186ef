"""numpy restatement of the reference's symbol kernels (TEST INFRASTRUCTURE ONLY).

Follows the *inference* path of the reference, i.e. the CUDA kernels in
/root/reference/src/layers/extensions/inference/elementwise/stream.cu, op by op and with one fp16
rounding per fp16 op exactly as they have it (numpy float16 arithmetic rounds each op to fp16):

  process_with_mask_kernel            stream.cu:549-630   (mask, subtract mean, round half away,
                                                           force-zero below skip threshold, clamp
                                                           to int8, add mean)
  single_part_for_writing/reading_4x  stream.cu:896-949   (fold the four channel groups)
  scale_to_index / build_index_*      stream.cu:77-161    (fp16 log-scale index, int16 packing)
  conditional_index_* / recover       stream.cu:176-383   (stream compaction and its inverse)
  restore_y_4x*                       stream.cu:757-844
  round_z_kernel                      stream.cu:862-884
  masks                               dmci_proxy.cpp:678-699 == common_model.py:174-195

All tensors are NHWC numpy float16 arrays [H, W, C]; symbols are flattened in NHWC order
(stream.cu:96-97).
"""
import numpy as np

F16 = np.float16

# def_const.h:6-12
SCALE_MIN = np.float32(0.11)
SCALE_MAX = np.float32(16.0)
LOG_SCALE_MIN = np.float32(-2.2073)
LOG_SCALE_MAX = np.float32(2.7726)
LOG_SCALE_STEP = np.float32((LOG_SCALE_MAX - LOG_SCALE_MIN) / np.float32(127))
LOG_SCALE_STEP_RECIP = np.float32(np.float32(1.0) / LOG_SCALE_STEP)


def round_half_away(x):
    """C round() on float32 (the kernels call round() on a float)."""
    x = x.astype(np.float32)
    return np.copysign(np.floor(np.abs(x) + np.float32(0.5)), x)


def get_mask_4x(H, W, C):
    """bool masks [4][H, W, C] (dmci_proxy.cpp:678-699)."""
    hh = np.arange(H)[:, None] & 1
    ww = np.arange(W)[None, :] & 1
    m = [(hh == 0) & (ww == 0), (hh == 0) & (ww == 1), (hh == 1) & (ww == 0), (hh == 1) & (ww == 1)]
    order = [(0, 1, 2, 3), (3, 2, 1, 0), (2, 3, 0, 1), (1, 0, 3, 2)]
    cq = C // 4
    masks = []
    for o in order:
        mk = np.zeros((H, W, C), dtype=bool)
        for g in range(4):
            mk[:, :, g * cq:(g + 1) * cq] = m[o[g]][:, :, None]
        masks.append(mk)
    return masks


def scale_to_index(scale):
    """stream.cu:77-87 in fp16: clamp, hlog (taken as correctly rounded), subtract, multiply;
    then __half2int_rd (floor)."""
    s = scale.astype(F16)
    s = np.where(s > F16(SCALE_MIN), s, F16(SCALE_MIN))     # max(scale, min); NaN/neg -> min
    s = np.where(s < F16(SCALE_MAX), s, F16(SCALE_MAX))
    lg = np.log(s.astype(np.float64)).astype(F16)
    d = (lg - F16(LOG_SCALE_MIN)).astype(F16)
    v = (d * F16(LOG_SCALE_STEP_RECIP)).astype(F16)
    return np.floor(v.astype(np.float32)).astype(np.int32)


def process_with_mask(y, scales, means, mask, thres):
    """-> y_q, y_hat, s_hat (all fp16 [H, W, C])."""
    thres = F16(np.float32(thres))
    zero = F16(0)
    s_hat = np.where(mask, scales, zero).astype(F16)
    means_hat = np.where(mask, means, zero).astype(F16)
    y_res = np.where(mask, (y - means_hat).astype(F16), zero).astype(F16)
    y_q = round_half_away(y_res).astype(F16)
    y_q = np.where(s_hat > thres, y_q, zero).astype(F16)
    y_q = np.maximum(np.minimum(y_q, F16(127)), F16(-128)).astype(F16)
    y_hat = (y_q + means_hat).astype(F16)
    return y_q, y_hat, s_hat


def fold4(x):
    """single_part_for_writing_4x: x1 + x2 + x3 + x4 over the channel groups (fp16 adds)."""
    cq = x.shape[-1] // 4
    a = (x[..., 0:cq] + x[..., cq:2 * cq]).astype(F16)
    a = (a + x[..., 2 * cq:3 * cq]).astype(F16)
    return (a + x[..., 3 * cq:]).astype(F16)


def build_index_enc(y_q_w, s_w, thres):
    """-> int16 (symbol << 8) + index, bool keep; flattened NHWC."""
    idx = scale_to_index(s_w)
    sym = np.floor(y_q_w.astype(np.float32)).astype(np.int32)
    comb = ((sym << 8) + idx).astype(np.int16)
    keep = s_w > F16(np.float32(thres))
    return comb.reshape(-1), keep.reshape(-1)


def build_index_dec(s_r, thres):
    idx = scale_to_index(s_r).astype(np.uint8)
    keep = s_r > F16(np.float32(thres))
    return idx.reshape(-1), keep.reshape(-1)


def recover(decoded_i8, keep, shape):
    out = np.zeros(keep.size, dtype=F16)
    out[keep] = decoded_i8.astype(F16)
    return out.reshape(shape)


def restore_y_4x(y_q_r, means, mask):
    """(y + means[group]) * mask per group (stream.cu:757-792)."""
    y4 = np.concatenate([y_q_r] * 4, axis=-1)
    return np.where(mask, (y4 + means).astype(F16), F16(0)).astype(F16)


def round_z(z):
    v = round_half_away(z)
    v = np.minimum(np.maximum(v, np.float32(-64)), np.float32(63))
    return v.astype(F16), v.astype(np.int8)


# ------------------------------------------------------------------ 2x checkerboard (inter LD / HT-S)
def get_mask_2x(H, W, C):
    """mask_0 = cat(m0, m1), mask_1 = cat(m1, m0) over the two channel halves
    (dmc_ld_proxy.cpp:672-683 == common_model.py:157-172)."""
    hh = np.arange(H)[:, None] & 1
    ww = np.arange(W)[None, :] & 1
    m0 = (hh == ww)
    m1 = ~m0
    half = C // 2
    mk0 = np.zeros((H, W, C), dtype=bool)
    mk1 = np.zeros((H, W, C), dtype=bool)
    mk0[:, :, :half] = m0[:, :, None]
    mk0[:, :, half:] = m1[:, :, None]
    mk1[:, :, :half] = m1[:, :, None]
    mk1[:, :, half:] = m0[:, :, None]
    return mk0, mk1


def clamp_min_half(q):
    """max(q, 0.5) in fp16 (stream.cu:60-61,437-438)."""
    return np.maximum(q.astype(F16), F16(0.5)).astype(F16)


def divide_with_clamp(y, q):
    """y * rcp(max(q, 0.5)), stream.cu:422-443; rcp taken as the correctly rounded fp16
    reciprocal, then one fp16 multiply."""
    r = (np.float32(1.0) / clamp_min_half(q).astype(np.float32)).astype(F16)
    return (y.astype(F16) * r).astype(F16)


def process_with_mask_2x(y, scales, means, mask, thres):
    """process_with_mask_kernel<scale_out=false> (stream.cu:549-630): y_q, y_hat."""
    y_q, y_hat, _ = process_with_mask(y, scales, means, mask, thres)
    return y_q, y_hat


def restore_y(y_q_r, means, mask):
    """restore_y_kernel (stream.cu:686-729): (y + means) * mask."""
    return np.where(mask, (y_q_r + means).astype(F16), F16(0)).astype(F16)
