// dwconv.hip - depthwise 3x3 convolution, NHWC fp16, zero padding, no bias
// (reference: cutlass/d3x3.cu:443-446, d3x3_kernel.h:103-166; the bias is folded into the next
// 1x1 conv by the host, layers_proxy.cpp:175-178).
//
// HBM-bound (2.9 of 949 GMAC per DMCI frame): each lane owns 8 channels (16 B) and a column of
// RPT output rows, sliding a 3-row register window so every input row is loaded once per lane
// column instead of three times; neighbouring columns come from L1/L2.
// fp32 accumulation in the fixed tap order (ky, kx) ascending, out-of-picture taps skipped.
#include "arith.h"
#include "ops.h"

namespace dcvc {

namespace {

constexpr int RPT = 4;   // output rows per thread

__global__ void __launch_bounds__(256)
dwconv3x3_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wt,
                 half_t* __restrict__ y, int ldy, int H, int W, int C)
{
    const int cv = C >> 3;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int hb_count = (H + RPT - 1) / RPT;
    if (i >= hb_count * W * cv) return;
    const int c0 = (i % cv) * 8;
    const int t = i / cv;
    const int w = t % W;
    const int h0 = (t / W) * RPT;

    float wgt[9][8];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const half8 w8 = *reinterpret_cast<const half8*>(wt + k * C + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) wgt[k][e] = static_cast<float>(w8[e]);
    }
    const bool left = w > 0, right = w + 1 < W;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int h = h0 + r;
        if (h >= H) break;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ih = h + ky - 1;
            if (ih < 0 || ih >= H) continue;
            const half_t* row = x + (static_cast<size_t>(ih) * W + w) * ldx + c0;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if ((kx == 0 && !left) || (kx == 2 && !right)) continue;
                const half8 v = *reinterpret_cast<const half8*>(row + (kx - 1) * ldx);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(static_cast<float>(v[e]), wgt[ky * 3 + kx][e], acc[e]);
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_half(acc[e]);
        *reinterpret_cast<half8*>(y + (static_cast<size_t>(h) * W + w) * ldy + c0) = o;
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
