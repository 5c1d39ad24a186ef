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
// Round 5 measured the load schedule of this walk (profiles/r05_dwconv_ab.txt; 136 x 240 pixels, us per launch for 384 / 128
// channels): the prediction was "latency-bound: one row of loads in flight per lane, ten dependent round trips per strip" -
// and it was wrong. All (rows + 2) x 3 loads of a 6-row strip issued in front of the first output row: 14.5 / 8.2 (compiler's
// register budget) and 15.5 / 8.6 (156 registers, 33 loads in front) against 12.9 / 6.7 for round 1's window: two of every
// three loads are neighbour columns that hit in L1 only while the lines of the lanes next door are still there, and ~ 300 KB
// in flight per CU against 32 KB of L1 turns those hits into L2 traffic. Strip height x look-ahead rows, same walk:
//   8,1  12.6 / 6.8    8,2  12.3 / 6.6    8,3  12.2 / 6.7    4,1  13.6 / 6.0    4,2  13.4 / 5.7    16,1  15.5 / 9.4    16,2  15.0 / 9.6
// i.e. look-ahead is worth 2 - 3 %, the strip height trades halo re-reads against waves per CU, and the launch moves its
// 50.1 MB at 4.0 - 4.2 TB/s back to back - 0.64 - 0.66 of the 6.29 TB/s a float4 copy reaches on this chip
// (MI355X_MICROARCH.md), 0.75 of it with the ~ 2 us launch boundary taken out. Default since: 8 rows / 2 ahead, 4 rows / 2
// ahead for <= 128 channels; DCVC_DWCONV_VARIANT = "<rows>,<ahead>" picks one instantiation for a process ("8,1" = round 1's
// kernel).
#include "arith.h"
#include "ops.h"

#include <cstdlib>
#include <stdexcept>
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

// the same walk with ROWS output rows per lane and AHEAD rows of loads in flight (AHEAD = 1: the kernel above)
template <int ROWS, int AHEAD>
__global__ void __launch_bounds__(256)
dwconv3x3_walk_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wt,
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

    half8 wgt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wgt[k] = *reinterpret_cast<const half8*>(wt + k * C + c0);
    const bool left = w > 0, right = w + 1 < W;
    const half8 zero = { 0, 0, 0, 0, 0, 0, 0, 0 };
    // ring of 2 + AHEAD rows: slot (r mod N) holds input row h0 - 1 + r
    constexpr int N = 2 + AHEAD;
    half8 win[N][3];
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
#pragma unroll
    for (int r = 0; r < N - 1; ++r) load_row(h0 - 1 + r, win[r]);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int h = h0 + r;
        if (h >= H) break;
        if (r + N - 1 < ROWS + 2) load_row(h0 - 1 + r + N - 1, win[(r + N - 1) % N]);
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
                    acc[e] = fmaf(static_cast<float>(win[(r + ky) % N][kx][e]), static_cast<float>(wgt[ky * 3 + kx][e]), acc[e]);
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_half(acc[e]);
        store_line(y + (static_cast<size_t>(h) * W + w) * ldy + c0, o);
    }
}

// DCVC_DWCONV_VARIANT = "<rows>,<ahead>"; -1 = unset: by channel count; 0 = the round-1 kernel (rows 8, one row ahead, shifting window)
int variant()
{
    static const int v = [] {
        const char* e = getenv("DCVC_DWCONV_VARIANT");
        const std::string s(e ? e : "");
        if (s.empty()) return -1;
        if (s == "8,2") return 1;
        if (s == "4,1") return 2;
        if (s == "4,2") return 3;
        if (s == "16,1") return 4;
        if (s == "16,2") return 5;
        if (s == "8,3") return 6;
        if (s == "8,1") return 0;
        // (a typo used to select round 1's kernel silently: advisor, round 5)
        throw std::invalid_argument("DCVC_DWCONV_VARIANT: unknown value '" + s + "' (8,1 | 8,2 | 8,3 | 4,1 | 4,2 | 16,1 | 16,2)");
    }();
    return v;
}

template <int ROWS, int AHEAD>
void launch_walk(const half_t* x, int ldx, const half_t* wt, half_t* y, int ldy, int H, int W, int C, hipStream_t stream)
{
    const long long n = static_cast<long long>((H + ROWS - 1) / ROWS) * W * (C / 8);
    hipLaunchKernelGGL((dwconv3x3_walk_kernel<ROWS, AHEAD>), dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                       stream, x, ldx, wt, y, ldy, H, W, C);
    hip_check(hipGetLastError(), "dwconv3x3 launch");
}

}  // namespace

void dwconv3x3(const half_t* x, int ldx, const half_t* wt, half_t* y, int ldy, int H, int W, int C,
               hipStream_t stream)
{
    if (C % 8 != 0) throw std::invalid_argument("dwconv3x3: C must be a multiple of 8");
    switch (variant() >= 0 ? variant() : (C <= 128 ? 3 : 1)) {
    case 1: return launch_walk<8, 2>(x, ldx, wt, y, ldy, H, W, C, stream);
    case 2: return launch_walk<4, 1>(x, ldx, wt, y, ldy, H, W, C, stream);
    case 3: return launch_walk<4, 2>(x, ldx, wt, y, ldy, H, W, C, stream);
    case 4: return launch_walk<16, 1>(x, ldx, wt, y, ldy, H, W, C, stream);
    case 5: return launch_walk<16, 2>(x, ldx, wt, y, ldy, H, W, C, stream);
    case 6: return launch_walk<8, 3>(x, ldx, wt, y, ldy, H, W, C, stream);
    default: break;
    }
    const long long n = static_cast<long long>((H + RPT - 1) / RPT) * W * (C / 8);
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                       stream, x, ldx, wt, y, ldy, H, W, C);
    hip_check(hipGetLastError(), "dwconv3x3 launch");
}

}  // namespace dcvc
