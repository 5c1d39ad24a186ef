// dwconv.hip - depthwise 3x3 convolution, NHWC fp16, zero padding, no bias
// (reference: cutlass/d3x3.cu:443-446, d3x3_kernel.h:103-166; the bias is folded into the next
// 1x1 conv by the host, layers_proxy.cpp:175-178).
//
// HBM-bound (2.9 of 949 GMAC per DMCI frame): each lane owns 8 channels (16 B) of one picture
// column and walks RPT output rows with a sliding 3x3 register window, so every input element is
// loaded once per lane column (plus the two halo rows per strip); the left / right neighbour
// columns are the adjacent lanes' lines in L1. fp32 accumulation in the fixed tap order (ky, kx)
// ascending, out-of-picture taps skipped (the oracle restates the same chain).
#include "arith.h"
#include "ops.h"

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

}  // namespace

void dwconv3x3(const half_t* x, int ldx, const half_t* wt, half_t* y, int ldy, int H, int W, int C,
               hipStream_t stream)
{
    if (C % 8 != 0) throw std::invalid_argument("dwconv3x3: C must be a multiple of 8");
    const long long n = static_cast<long long>((H + RPT - 1) / RPT) * W * (C / 8);
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                       stream, x, ldx, wt, y, ldy, H, W, C);
    hip_check(hipGetLastError(), "dwconv3x3 launch");
}

}  // namespace dcvc
