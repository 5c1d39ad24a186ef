// dwconv.hip - depthwise 3x3 convolution, NHWC fp16, zero padding, no bias
// (reference: cutlass/d3x3.cu:443-446, d3x3_kernel.h:103-166; the bias is folded into the next
// 1x1 conv by the host, layers_proxy.cpp:175-178).
//
// HBM-bound (2.9 of 949 GMAC per DMCI frame): each lane owns 8 channels (16 B) of one picture
// column and RPT output rows, so every input element is loaded once per lane column (plus the two
// halo rows per strip); the left / right neighbour columns are the adjacent lanes' lines in L1.
// fp32 accumulation in the fixed tap order (ky, kx) ascending, out-of-picture taps skipped (the
// oracle restates the same chain).
//
// Round 5: ALL (RPT + 2) x 3 loads of a lane are issued before the first output row is computed.
// Round 1's kernel walked the rows with a sliding window - load row h + 1, wait, 72 fmaf, store,
// next row - i.e. ONE row of loads in flight per wave and a serial chain of RPT + 2 memory round
// trips; at 12 waves per CU with two thirds of the loads L1 hits that is ~ 12 KB of distinct bytes
// in flight per CU, and by Little's law (~ 1 us to the Infinity Cache / HBM) ~ 3 TB/s for the chip:
// exactly what the launch measured (41.2 MB in 13.4 us = 0.38 of the HBM peak).
// With the loads up front the counted waits let row r's arithmetic start when rows <= r + 1 have
// arrived while everything behind them is still in flight. DCVC_DWCONV_MODE=sliding selects the old
// kernel (A/B partner, same arithmetic).
#include "arith.h"
#include "ops.h"

#include <cstdlib>
#include <string>

namespace dcvc {

namespace {

constexpr int RPT = 8;   // output rows per thread

__global__ void __launch_bounds__(256)
dwconv3x3_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wt,
                 half_t* __restrict__ y, int ldy, int H, int W, int C)
{
    const int cv = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int strips = (H + RPT - 1) / RPT;
    if (i >= strips * W * cv) return;
    const int c0 = (i % cv) * 8;
    const int t = i / cv;
    const int w = t % W;
    const int h0 = (t / W) * RPT;

    half8 wgt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wgt[k] = *reinterpret_cast<const half8*>(wt + k * C + c0);
    const bool left = w > 0, right = w + 1 < W;
    const half8 zero = { 0, 0, 0, 0, 0, 0, 0, 0 };

    // win[r][c]: input rows h-1, h, h+1 (r) x columns w-1, w, w+1 (c)
    half8 win[3][3];
    auto load_row = [&](int ih, half8 (&dst)[3]) {
        if (ih < 0 || ih >= H) {
            dst[0] = dst[1] = dst[2] = zero;
            return;
        }
        const half_t* row = x + (static_cast<size_t>(ih) * W + w) * ldx + c0;
        dst[1] = *reinterpret_cast<const half8*>(row);
        dst[0] = left ? *reinterpret_cast<const half8*>(row - ldx) : zero;
        dst[2] = right ? *reinterpret_cast<const half8*>(row + ldx) : zero;
    };
    load_row(h0 - 1, win[0]);
    load_row(h0, win[1]);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int h = h0 + r;
        if (h >= H) break;
        load_row(h + 1, win[2]);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ih = h + ky - 1;
            if (ih < 0 || ih >= H) continue;          // skipped taps: no fmaf at all
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if ((kx == 0 && !left) || (kx == 2 && !right)) continue;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] = fmaf(static_cast<float>(win[ky][kx][e]), static_cast<float>(wgt[ky * 3 + kx][e]), acc[e]);
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_half(acc[e]);
        store_line(y + (static_cast<size_t>(h) * W + w) * ldy + c0, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            win[0][c] = win[1][c];
            win[1][c] = win[2][c];
        }
    }
}

// the round-5 form: (ROWS + 2) x 3 x 16 B in flight per lane, then the same arithmetic in the same order.
// Lanes whose 3 x 3 windows all lie inside the picture (90 % at 1080p / 8) run straight-line code with every load
// unconditional and unconditionally used - otherwise the compiler sinks the loads of a conditionally skipped row into the
// branch that uses them and waits for ALL loads in front of the first fmaf (seen in the ISA of a first version); lanes at
// the picture border take the sliding-window walk of the round-1 kernel (skipped taps: no fmaf at all).
template <int ROWS>
__device__ __forceinline__ void strip_interior(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wt,
                                               half_t* __restrict__ y, int ldy, int W, int C, int c0, int w, int h0)
{
    // the weights FIRST: they are the oldest loads then, and the counted wait in front of row 0's arithmetic covers them
    // and the first three input rows only
    half8 wgt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wgt[k] = *reinterpret_cast<const half8*>(wt + k * C + c0);
    half8 in[ROWS + 2][3];
#pragma unroll
    for (int r = 0; r < ROWS + 2; ++r) {
        const half_t* row = x + (static_cast<size_t>(h0 - 1 + r) * W + w) * ldx + c0;
        in[r][0] = *reinterpret_cast<const half8*>(row - ldx);
        in[r][1] = *reinterpret_cast<const half8*>(row);
        in[r][2] = *reinterpret_cast<const half8*>(row + ldx);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] = fmaf(static_cast<float>(in[r + ky][kx][e]), static_cast<float>(wgt[ky * 3 + kx][e]), acc[e]);
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_half(acc[e]);
        store_line(y + (static_cast<size_t>(h0 + r) * W + w) * ldy + c0, o);
    }
}

template <int ROWS>
__device__ __forceinline__ void strip_border(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wt,
                                             half_t* __restrict__ y, int ldy, int H, int W, int C, int c0, int w, int h0)
{
    half8 wgt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wgt[k] = *reinterpret_cast<const half8*>(wt + k * C + c0);
    const bool left = w > 0, right = w + 1 < W;
    const half8 zero = { 0, 0, 0, 0, 0, 0, 0, 0 };
    half8 win[3][3];
    auto load_row = [&](int ih, half8 (&dst)[3]) {
        if (ih < 0 || ih >= H) {
            dst[0] = dst[1] = dst[2] = zero;
            return;
        }
        const half_t* row = x + (static_cast<size_t>(ih) * W + w) * ldx + c0;
        dst[1] = *reinterpret_cast<const half8*>(row);
        dst[0] = left ? *reinterpret_cast<const half8*>(row - ldx) : zero;
        dst[2] = right ? *reinterpret_cast<const half8*>(row + ldx) : zero;
    };
    load_row(h0 - 1, win[0]);
    load_row(h0, win[1]);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int h = h0 + r;
        if (h >= H) break;
        load_row(h + 1, win[2]);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ih = h + ky - 1;
            if (ih < 0 || ih >= H) continue;          // skipped taps: no fmaf at all
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if ((kx == 0 && !left) || (kx == 2 && !right)) continue;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] = fmaf(static_cast<float>(win[ky][kx][e]), static_cast<float>(wgt[ky * 3 + kx][e]), acc[e]);
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_half(acc[e]);
        store_line(y + (static_cast<size_t>(h) * W + w) * ldy + c0, o);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            win[0][c] = win[1][c];
            win[1][c] = win[2][c];
        }
    }
}

template <int ROWS>
__device__ __forceinline__ void ahead_body(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wt,
                                           half_t* __restrict__ y, int ldy, int H, int W, int C)
{
    const int cv = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int strips = (H + ROWS - 1) / ROWS;
    if (i >= strips * W * cv) return;
    const int c0 = (i % cv) * 8;
    const int t = i / cv;
    const int w = t % W;
    const int h0 = (t / W) * ROWS;
    if (h0 >= 1 && h0 + ROWS + 1 <= H && w >= 1 && w + 1 < W) {
        strip_interior<ROWS>(x, ldx, wt, y, ldy, W, C, c0, w, h0);
    } else {
        strip_border<ROWS>(x, ldx, wt, y, ldy, H, W, C, c0, w, h0);
    }
}

// two register budgets of the same body: the compiler's default (99 registers, 4 - 5 waves per SIMD; its scheduler then keeps
// ~ 16 loads in front of the first row and fetches the rest one row ahead) and 3 waves per SIMD (156 registers: 21 loads in
// front of row 0, the other 12 right behind it). DCVC_DWCONV_MODE = ahead | deep | sliding picks one per process.
template <int ROWS>
__global__ void __launch_bounds__(256)
dwconv3x3_ahead_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wt,
                       half_t* __restrict__ y, int ldy, int H, int W, int C)
{
    ahead_body<ROWS>(x, ldx, wt, y, ldy, H, W, C);
}

template <int ROWS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 3)))
dwconv3x3_deep_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wt,
                      half_t* __restrict__ y, int ldy, int H, int W, int C)
{
    ahead_body<ROWS>(x, ldx, wt, y, ldy, H, W, C);
}

constexpr int ROWS_AHEAD = 6;     // 8 x 3 loads = 96 registers of input per lane

enum Mode : int { kSliding = 0, kAhead = 1, kDeep = 2 };

Mode mode()
{
    static const Mode m = [] {
        const char* e = getenv("DCVC_DWCONV_MODE");
        const std::string v(e ? e : "");
        if (const char* old = getenv("DCVC_DWCONV_SLIDING")) {
            if (old[0] == '1') return kSliding;
        }
        return v == "sliding" ? kSliding : v == "ahead" ? kAhead : kDeep;
    }();
    return m;
}

}  // namespace

void dwconv3x3(const half_t* x, int ldx, const half_t* wt, half_t* y, int ldy, int H, int W, int C,
               hipStream_t stream)
{
    if (C % 8 != 0) throw std::invalid_argument("dwconv3x3: C must be a multiple of 8");
    if (mode() != kSliding) {
        const long long n = static_cast<long long>((H + ROWS_AHEAD - 1) / ROWS_AHEAD) * W * (C / 8);
        const dim3 grid(static_cast<unsigned>((n + 255) / 256));
        if (mode() == kDeep) {
            hipLaunchKernelGGL(dwconv3x3_deep_kernel<ROWS_AHEAD>, grid, dim3(256), 0, stream, x, ldx, wt, y, ldy, H, W, C);
        } else {
            hipLaunchKernelGGL(dwconv3x3_ahead_kernel<ROWS_AHEAD>, grid, dim3(256), 0, stream, x, ldx, wt, y, ldy, H, W, C);
        }
        hip_check(hipGetLastError(), "dwconv3x3 launch");
        return;
    }
    const long long n = static_cast<long long>((H + RPT - 1) / RPT) * W * (C / 8);
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                       stream, x, ldx, wt, y, ldy, H, W, C);
    hip_check(hipGetLastError(), "dwconv3x3 launch");
}

}  // namespace dcvc
