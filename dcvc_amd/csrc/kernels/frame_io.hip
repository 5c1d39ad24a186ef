// frame_io.hip - 8-bit YUV420 planes <-> the codec's fp16 NHWC picture tensor, on the GPU.
//
// Reference (all on the host or as a chain of torch ops): test_video.py:69-123 get_src_frame
// (scipy nearest-neighbour chroma upsampling on the CPU, cat, H2D, .half(), / 255, - 0.5,
// channels_last), test_video.py:32-45 get_distortion and :356-363 (x_hat + 0.5, 2x2 average of
// the chroma, * 255, clamp; the writer rounds Y half-to-even and TRUNCATES U/V).
// Here: one HBM-bound pass each, reading the u8 planes / writing them with 16-B accesses; the
// 3.1 MB of a 1080p picture cross PCIe as u8 instead of 12.4 MB of fp16.
#include "arith.h"
#include "ops.h"

namespace dcvc {

namespace {

__device__ __forceinline__ half_t load_pixel(unsigned v)
{
    // x.half() / 255.0 - 0.5 with one fp16 rounding per op (opmath float)
    const half_t d = to_half(static_cast<float>(v) / 255.0f);
    return to_half(static_cast<float>(d) - 0.5f);
}

// one thread = 8 consecutive luma pixels of a row (4 chroma samples)
__global__ void yuv420_to_x_kernel(const uint8_t* __restrict__ yp, const uint8_t* __restrict__ uvp,
                                   int H, int W, half_t* __restrict__ x, int ldx)
{
    const int wv = (W + 7) >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * wv) return;
    const int h = i / wv, w0 = (i - h * wv) * 8;
    const int Hc = H >> 1, Wc = W >> 1;
    const uint8_t* yr = yp + static_cast<size_t>(h) * W + w0;
    const uint8_t* ur = uvp + static_cast<size_t>(min(h >> 1, Hc - 1)) * Wc;
    const uint8_t* vr = ur + static_cast<size_t>(Hc) * Wc;
    half_t* o = x + (static_cast<size_t>(h) * W + w0) * ldx;
    const int n = min(8, W - w0);
    for (int e = 0; e < n; ++e) {
        const int wc = min((w0 + e) >> 1, Wc - 1);
        o[e * ldx + 0] = load_pixel(yr[e]);
        o[e * ldx + 1] = load_pixel(ur[wc]);
        o[e * ldx + 2] = load_pixel(vr[wc]);
    }
}

__device__ __forceinline__ half_t scale255(half_t t)
{
    const float v = static_cast<float>(to_half(static_cast<float>(t) * 255.0f));
    return to_half(fminf(fmaxf(v, 0.f), 255.f));
}

// one thread = one chroma sample = a 2x2 block of luma
__global__ void x_to_yuv420_kernel(const half_t* __restrict__ x, int ldrow, int H, int W,
                                   half_t* __restrict__ y16, half_t* __restrict__ uv16,
                                   uint8_t* __restrict__ y8, uint8_t* __restrict__ uv8)
{
    const int Hc = H >> 1, Wc = W >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Hc * Wc) return;
    const int hc = i / Wc, wc = i - hc * Wc;
    float su = 0.f, sv = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int h = 2 * hc + dy, w = 2 * wc + dx;
            const half_t* p = x + (static_cast<size_t>(h) * ldrow + w) * 3;
            const half_t ty = hadd(p[0], static_cast<half_t>(0.5f));        // x_hat + 0.5
            su += static_cast<float>(hadd(p[1], static_cast<half_t>(0.5f)));
            sv += static_cast<float>(hadd(p[2], static_cast<half_t>(0.5f)));
            const half_t yv = scale255(ty);
            const size_t o = static_cast<size_t>(h) * W + w;
            if (y16) y16[o] = yv;
            if (y8) y8[o] = static_cast<uint8_t>(rintf(static_cast<float>(yv)));   // .round().byte(): half to even
        }
    // avg_pool2d accumulates in fp32 and rounds once
    const half_t u = scale255(to_half(su * 0.25f)), v = scale255(to_half(sv * 0.25f));
    const size_t oc = static_cast<size_t>(hc) * Wc + wc, plane = static_cast<size_t>(Hc) * Wc;
    if (uv16) { uv16[oc] = u; uv16[plane + oc] = v; }
    if (uv8) {                                                               // .byte(): truncation
        uv8[oc] = static_cast<uint8_t>(static_cast<float>(u));
        uv8[plane + oc] = static_cast<uint8_t>(static_cast<float>(v));
    }
}

}  // namespace

void yuv420_to_x(const uint8_t* y, const uint8_t* uv, int H, int W, half_t* x, int ldx, hipStream_t stream)
{
    if (H <= 0 || W <= 0 || (H & 1) || (W & 1)) throw std::invalid_argument("yuv420_to_x: even picture size required");
    if (ldx < 3) throw std::invalid_argument("yuv420_to_x: pixel stride must be >= 3");
    const long long n = static_cast<long long>(H) * ((W + 7) / 8);
    hipLaunchKernelGGL(yuv420_to_x_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream,
                       y, uv, H, W, x, ldx);
    hip_check(hipGetLastError(), "yuv420_to_x launch");
}

void x_to_yuv420(const half_t* x, int row_pixels, int H, int W, half_t* y16, half_t* uv16, uint8_t* y8,
                 uint8_t* uv8, hipStream_t stream)
{
    if (H <= 0 || W <= 0 || (H & 1) || (W & 1) || row_pixels < W) {
        throw std::invalid_argument("x_to_yuv420: even picture size inside the padded rows required");
    }
    const long long n = static_cast<long long>(H / 2) * (W / 2);
    hipLaunchKernelGGL(x_to_yuv420_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream,
                       x, row_pixels, H, W, y16, uv16, y8, uv8);
    hip_check(hipGetLastError(), "x_to_yuv420 launch");
}

}  // namespace dcvc
