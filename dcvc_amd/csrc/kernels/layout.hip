// layout.hip - HBM-bound data-movement kernels: pad + pixel-unshuffle(8) of the source picture,
// pixel-shuffle(8)+clamp of the reconstruction, pixel-shuffle(2), replicate pad / crop of the
// latent, per-channel scale. Reference: elementwise/cat_and_pad.cu, elementwise/shuffle.cu,
// elementwise/stream.cu:485-547.
//
// All tensors are NHWC ("channels_last"); every kernel moves 16 bytes per lane on the
// contiguous (channel) axis of the WIDE side of the transform, so the 192-channel feature rows
// are written / read as full 384-byte lines.
#include "arith.h"
#include "ops.h"

namespace dcvc {

namespace {

// out[h8][w8][c*64 + dy*8 + dx] = x[min(h8*8+dy, H-1)][min(w8*8+dx, W-1)][c]
// one thread = one (pixel, c, dy): 8 consecutive dx -> 8 consecutive output channels (16 B store)
__global__ void pad_unshuffle8_kernel(const half_t* __restrict__ x, int H, int W, int C3,
                                      half_t* __restrict__ out, int H8, int W8, int ldout)
{
    const int per_pix = C3 * 8;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H8 * W8 * per_pix) return;
    const int pix = i / per_pix;
    const int r = i - pix * per_pix;
    const int c = r >> 3, dy = r & 7;
    const int h8 = pix / W8, w8 = pix - h8 * W8;
    const int sh = min(h8 * 8 + dy, H - 1);
    half8 v;
#pragma unroll
    for (int dx = 0; dx < 8; ++dx) {
        const int sw = min(w8 * 8 + dx, W - 1);
        v[dx] = x[(static_cast<size_t>(sh) * W + sw) * C3 + c];
    }
    *reinterpret_cast<half8*>(out + static_cast<size_t>(pix) * ldout + c * 64 + dy * 8) = v;
}

// out[h8*8+dy][w8*8+dx][c] = clamp(in[h8][w8][c*64 + dy*8 + dx])
template <bool CLAMP>
__global__ void shuffle8_kernel(const half_t* __restrict__ in, int ldin, int H8, int W8, int C3,
                                half_t* __restrict__ out)
{
    const int per_pix = C3 * 8;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H8 * W8 * per_pix) return;
    const int pix = i / per_pix;
    const int r = i - pix * per_pix;
    const int c = r >> 3, dy = r & 7;
    const int h8 = pix / W8, w8 = pix - h8 * W8;
    const half8 v = *reinterpret_cast<const half8*>(in + static_cast<size_t>(pix) * ldin + c * 64 + dy * 8);
    const int Wo = W8 * 8;
    half_t* o = out + (static_cast<size_t>(h8 * 8 + dy) * Wo + w8 * 8) * C3 + c;
#pragma unroll
    for (int dx = 0; dx < 8; ++dx) {
        float f = static_cast<float>(v[dx]);
        if (CLAMP) f = fminf(fmaxf(f, -0.5f), 0.5f);
        o[dx * C3] = to_half(f);
    }
}

// out[2h+i][2w+j][c] = in[h][w][c*4 + i*2 + j]; one thread = (out pixel, 8 channels)
__global__ void shuffle2_kernel(const half_t* __restrict__ in, int ldin, int H, int W, int C,
                                half_t* __restrict__ out, int ldout)
{
    const int cv = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * H * W * cv) return;
    const int opix = i / cv;
    const int c0 = (i - opix * cv) * 8;
    const int oh = opix / (2 * W), ow = opix - oh * (2 * W);
    const int sub = (oh & 1) * 2 + (ow & 1);
    const half_t* src = in + (static_cast<size_t>(oh >> 1) * W + (ow >> 1)) * ldin + c0 * 4 + sub;
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e * 4];
    *reinterpret_cast<half8*>(out + static_cast<size_t>(opix) * ldout + c0) = v;
}

__global__ void replicate_pad_kernel(const half_t* __restrict__ in, int ldin, int H, int W, int C,
                                     int Ho, int Wo, half_t* __restrict__ out, int ldout)
{
    const int cv = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ho * Wo * cv) return;
    const int opix = i / cv;
    const int c0 = (i - opix * cv) * 8;
    const int oh = opix / Wo, ow = opix - oh * Wo;
    const int sh = min(oh, H - 1), sw = min(ow, W - 1);
    *reinterpret_cast<half8*>(out + static_cast<size_t>(opix) * ldout + c0) =
        *reinterpret_cast<const half8*>(in + (static_cast<size_t>(sh) * W + sw) * ldin + c0);
}

__global__ void crop_kernel(const half_t* __restrict__ in, int ldin, int Win,
                            half_t* __restrict__ out, int ldout, int H, int W, int C)
{
    const int cv = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W * cv) return;
    const int opix = i / cv;
    const int c0 = (i - opix * cv) * 8;
    const int oh = opix / W, ow = opix - oh * W;
    *reinterpret_cast<half8*>(out + static_cast<size_t>(opix) * ldout + c0) =
        *reinterpret_cast<const half8*>(in + (static_cast<size_t>(oh) * Win + ow) * ldin + c0);
}

__global__ void mul_channel_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ q,
                                   half_t* __restrict__ y, int ldy, int pixels, int C)
{
    const int cv = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixels * cv) return;
    const int pix = i / cv;
    const int c0 = (i - pix * cv) * 8;
    const half8 v = *reinterpret_cast<const half8*>(x + static_cast<size_t>(pix) * ldx + c0);
    const half8 s = *reinterpret_cast<const half8*>(q + c0);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = hmul(v[e], s[e]);
    *reinterpret_cast<half8*>(y + static_cast<size_t>(pix) * ldy + c0) = o;
}

// y = x * max(q, 0.5)  or  y = x * fp16(1 / max(q, 0.5)), q a tensor of the same shape
template <bool RECIP>
__global__ void scale_clamped_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ q, int ldq,
                                     half_t* __restrict__ y, int ldy, int pixels, int C)
{
    const int cv = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixels * cv) return;
    const int pix = i / cv;
    const int c0 = (i - pix * cv) * 8;
    const half8 v = *reinterpret_cast<const half8*>(x + static_cast<size_t>(pix) * ldx + c0);
    const half8 s = *reinterpret_cast<const half8*>(q + static_cast<size_t>(pix) * ldq + c0);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        half_t qc = static_cast<float>(s[e]) > 0.5f ? s[e] : static_cast<half_t>(0.5f);
        if (RECIP) qc = to_half(1.0f / static_cast<float>(qc));
        o[e] = hmul(v[e], qc);
    }
    *reinterpret_cast<half8*>(y + static_cast<size_t>(pix) * ldy + c0) = o;
}

inline dim3 grid1d(long long n, int block = 256)
{
    return dim3(static_cast<unsigned>((n + block - 1) / block));
}

}  // namespace

void pad_unshuffle8(const half_t* x, int H, int W, int C3, half_t* out, int H8, int W8,
                    hipStream_t stream, int ldout)
{
    if (ldout == 0) ldout = C3 * 64;
    const long long n = static_cast<long long>(H8) * W8 * C3 * 8;
    hipLaunchKernelGGL(pad_unshuffle8_kernel, grid1d(n), dim3(256), 0, stream, x, H, W, C3, out, H8, W8, ldout);
    hip_check(hipGetLastError(), "pad_unshuffle8 launch");
}

void shuffle8(const half_t* in, int ldin, int H8, int W8, int C3, bool clamp, half_t* out,
              hipStream_t stream)
{
    const long long n = static_cast<long long>(H8) * W8 * C3 * 8;
    if (clamp) hipLaunchKernelGGL(shuffle8_kernel<true>, grid1d(n), dim3(256), 0, stream, in, ldin, H8, W8, C3, out);
    else       hipLaunchKernelGGL(shuffle8_kernel<false>, grid1d(n), dim3(256), 0, stream, in, ldin, H8, W8, C3, out);
    hip_check(hipGetLastError(), "shuffle8 launch");
}

void shuffle2(const half_t* in, int ldin, int H, int W, int C, half_t* out, int ldout, hipStream_t stream)
{
    const long long n = 4LL * H * W * (C / 8);
    hipLaunchKernelGGL(shuffle2_kernel, grid1d(n), dim3(256), 0, stream, in, ldin, H, W, C, out, ldout);
    hip_check(hipGetLastError(), "shuffle2 launch");
}

void replicate_pad(const half_t* in, int ldin, int H, int W, int C, int pad_b, int pad_r,
                   half_t* out, int ldout, hipStream_t stream)
{
    const int Ho = H + pad_b, Wo = W + pad_r;
    const long long n = static_cast<long long>(Ho) * Wo * (C / 8);
    hipLaunchKernelGGL(replicate_pad_kernel, grid1d(n), dim3(256), 0, stream, in, ldin, H, W, C, Ho, Wo, out, ldout);
    hip_check(hipGetLastError(), "replicate_pad launch");
}

void crop(const half_t* in, int ldin, int Win, half_t* out, int ldout, int H, int W, int C,
          hipStream_t stream)
{
    const long long n = static_cast<long long>(H) * W * (C / 8);
    hipLaunchKernelGGL(crop_kernel, grid1d(n), dim3(256), 0, stream, in, ldin, Win, out, ldout, H, W, C);
    hip_check(hipGetLastError(), "crop launch");
}

void mul_channel(const half_t* x, int ldx, const half_t* q, half_t* y, int ldy, int pixels, int C,
                 hipStream_t stream)
{
    const long long n = static_cast<long long>(pixels) * (C / 8);
    hipLaunchKernelGGL(mul_channel_kernel, grid1d(n), dim3(256), 0, stream, x, ldx, q, y, ldy, pixels, C);
    hip_check(hipGetLastError(), "mul_channel launch");
}

void scale_clamped(const half_t* x, int ldx, const half_t* q, int ldq, half_t* y, int ldy, int pixels,
                   int C, bool reciprocal, hipStream_t stream)
{
    if (C % 8 != 0) throw std::invalid_argument("scale_clamped: C must be a multiple of 8");
    const long long n = static_cast<long long>(pixels) * (C / 8);
    if (reciprocal) hipLaunchKernelGGL(scale_clamped_kernel<true>, grid1d(n), dim3(256), 0, stream, x, ldx, q, ldq, y, ldy, pixels, C);
    else            hipLaunchKernelGGL(scale_clamped_kernel<false>, grid1d(n), dim3(256), 0, stream, x, ldx, q, ldq, y, ldy, pixels, C);
    hip_check(hipGetLastError(), "scale_clamped launch");
}

}  // namespace dcvc
