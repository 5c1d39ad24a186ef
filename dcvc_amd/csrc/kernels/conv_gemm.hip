// conv_gemm.hip - the contraction kernel of the codec: every dense convolution of DCVC-UF
// (1x1, 2x2 stride 2, 3x3 stride 1/2, and the 2x2 stride-2 transposed conv) as one NHWC
// implicit GEMM on the gfx950 matrix cores with the fused epilogues of the reference's ten
// CUTLASS entry points:
//
//   conv1x1_bias                    conv1x1_bias.cu:524-540
//   conv1x1_bias_wsilu              conv1x1_bias_wsilu.cu:230-248
//   conv1x1_bias_shortcut           conv1x1_bias_shortcut.cu:230-247
//   conv1x1_bias_shortcut2          conv1x1_bias_shortcut2.cu:81-100
//   conv1x1_bias_shortcut_with_quant / conv1x1_bias_with_quant   (…_with_quant.cu)
//   conv1x1_bias_wsilu_chunk_add    conv1x1_bias_wsilu_chunk_add.cu:356-390
//   conv_bias (k in {2,3}, stride in {1,2})   conv_bias.cu:131-151
//   transposed_conv (2x2, stride 2)           transposed_conv.cu:101-119
//
// Design (MI355X-first, not a CUTLASS translation):
//   * D^T = W * X^T: the weight fragment is the MFMA "A" operand and the activation fragment
//     the "B" operand of v_mfma_f32_32x32x16_f16, so a lane's 16 accumulators are 4 groups of 4
//     CONSECUTIVE output channels of ONE pixel. chunk-add (sum of 4 adjacent channels) is then a
//     purely in-register reduction and the NHWC store is channel-contiguous.
//   * block tile (WM*MT*32 pixels) x (WN*NT*32 channels) x 64 k, WM x WN waves each owning
//     MT x NT MFMA tiles; fp32 accumulation, k ascending in steps of 16 (fixed order -> results
//     do not depend on the tile shape; the oracle restates the same order). Shapes in use:
//     128x128 (4 waves), 64x128 (4 waves, small pictures / P16 layers), chosen per launch.
//   * global -> LDS staging by global_load_lds (16 B per lane, no VGPR round trip), two LDS
//     stages, one barrier per k-step. The LDS image is lane-linear, so the bank-conflict swizzle
//     (16-B chunk index XOR ((row >> 1) & 7)) is applied to the per-lane SOURCE address and to
//     the ds_read_b128 fragment address.
//   * every operand is (pointer, leading dimension): channel-slice views of wider NHWC buffers
//     are first-class (the reference's free torch.cat, conv1x1_kernel.h:82,102-107).
//   * epilogue straight from the accumulators: v_permlane32_swap pairs the two half-waves so
//     each lane owns 8 consecutive channels -> 16-B bias / residual loads and 16-B stores.
//     WSiLU is a 4 KiB piecewise-cubic table in LDS (arith.h).
//   * XCD-aware block order: consecutive logical tiles (same pixel rows, adjacent channel tiles)
//     run on the same XCD and share the activation tile in that XCD's L2.
#include "arith.h"
#include "ops.h"
#include "wsilu_table.h"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace dcvc {

namespace {

constexpr int BK = 64;             // k per stage
constexpr int WSILU_TABLE_BYTES = WSILU_SEGMENTS * 16;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct ConvGemmParams {
    const half_t* x;      // activations, pixel stride ldx
    const half_t* w;      // [N][K] (k contiguous; for k x k convs K = taps * Cin, tap major)
    const half_t* bias;   // [N] or nullptr
    const half_t* r1;     // residual 1 (pixel stride ldr1) or nullptr
    const half_t* r2;     // residual 2 or nullptr
    const half_t* q;      // per-channel scale [Nout] (fused "with_quant") or nullptr
    const half_t* q2;     // second per-channel scale applied to the ROUNDED output (models the
                          // reference's separate multiply_with_broadcast kernel) or nullptr
    const half_t* zeros;  // >= 128 B of zeros (padding taps)
    const float4* wsilu;  // WSiLU coefficient table (device memory)
    half_t* y;            // output, pixel stride ldy
    int ldx, ldr1, ldr2, ldy;
    int M, N, K;          // output pixels, output channels (pre chunk-add), contraction length
    // spatial description (k x k convs and the transposed conv)
    int in_h, in_w, out_h, out_w;   // input / output grid
    int cin, ksize, stride, pad;
    int up_cout;                    // transposed conv: N = 4 * up_cout, channel tile n0 belongs to output pixel
                                    // (2y + dy, 2x + dx) with dy * 2 + dx = n0 / up_cout (a tile never straddles two of them)
    long long* timeline;            // optional [blocks][16] shader-clock stamps of wave 0 (tools/gemm_timeline.py)
};

enum : int { ACT_NONE = 0, ACT_WSILU = 1 };

// ---------------------------------------------------------------------------------------------
template <bool SPATIAL>
__device__ __forceinline__ const half_t* x_row_ptr(const ConvGemmParams& p, int m, int k0)
{
    if constexpr (!SPATIAL) {
        return p.x + static_cast<size_t>(m) * p.ldx + k0;
    } else {
        const int tap = k0 / p.cin;
        const int c0 = k0 - tap * p.cin;
        const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
        const int oy = m / p.out_w, ox = m - oy * p.out_w;
        const int iy = oy * p.stride + ky - p.pad;
        const int ix = ox * p.stride + kx - p.pad;
        if (iy < 0 || iy >= p.in_h || ix < 0 || ix >= p.in_w) {
            return p.zeros;
        }
        return p.x + (static_cast<size_t>(iy) * p.in_w + ix) * p.ldx + c0;
    }
}

// XDIRECT: the activation operand does not pass through LDS. Each wave owns 32 pixel rows and ALL
// channels of the tile (WN = 1), so no other wave needs its activation fragments: a lane loads
// the 8 consecutive k values of its pixel straight from global memory into the MFMA operand
// registers (16 B per lane and k-slice, one 128-B line per row and k-step). Only the weight tile
// is staged in LDS - half the LDS fill traffic, all of it L2-resident weights - and the block
// needs 64 KB instead of 128 KB of LDS, so two blocks share a CU and the epilogue of one overlaps
// the main loop of the other.
template <int WM, int WN, int MT, int NT, int STAGES, bool SPATIAL, int ACT, bool CHUNK, int NRES, bool QUANT, bool UPSAMPLE,
          bool XDIRECT = false>
__global__ void __launch_bounds__(WM * WN * 64, XDIRECT ? 2 : 1)
conv_gemm_kernel(const ConvGemmParams p)
{
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int BM = WM * MT * 32;
    constexpr int BN = WN * NT * 32;
    constexpr int XT_BYTES = XDIRECT ? 0 : BM * BK * 2;
    constexpr int WT_BYTES = BN * BK * 2;
    constexpr int STAGE_BYTES = XT_BYTES + WT_BYTES;
    constexpr int XU = XDIRECT ? 0 : BM * 8 / NTHREADS;      // 16-B units per thread and stage
    constexpr int WU = BN * 8 / NTHREADS;
    static_assert(!XDIRECT || (WN == 1 && MT == 1 && STAGES == 2), "direct activation loads: one 32-row strip per wave");
    constexpr int ROWS_PER_PASS = NTHREADS / 8;
    static_assert(ROWS_PER_PASS % 16 == 0, "swizzle term must not depend on the pass");
    static_assert(!CHUNK || NT % 2 == 0, "chunk-add pairs two channel tiles");
    static_assert(STAGES == 2 || STAGES == 3, "two or three LDS stages");
    constexpr int BNO = CHUNK ? BN / 4 : BN;             // channels per output row of the tile
    constexpr int OCH = BNO / 8;                         // 16-B chunks per output row
    constexpr int OUNITS = BM * OCH / NTHREADS;          // output chunks per thread
    static_assert(BM * OCH % NTHREADS == 0 && BM * BNO * 2 <= STAGES * STAGE_BYTES, "output tile must fit");

    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (X tile, W tile) [+ table]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int stamp_no = 0;
    auto stamp = [&]() {
        if (p.timeline != nullptr && tid == 0 && stamp_no < 16) {
            p.timeline[static_cast<size_t>(blockIdx.x) * 16 + stamp_no] = static_cast<long long>(__builtin_readcyclecounter());
        }
        ++stamp_no;
    };
    stamp();                                                       // 0: kernel entry
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5;

    // ---- XCD-aware tile order (bijective chunking: XCD x gets a contiguous range of tiles)
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int tile;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int qn = nwg >> 3, rn = nwg & 7;
        tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
    }
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;

    // ---- staging plan: thread t moves 16-B unit u = j*NTHREADS + t of each tile
    //      unit u -> row u>>3, physical chunk u&7; logical chunk = physical ^ ((row>>1)&7)
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
    int xrow[XU > 0 ? XU : 1];
    const half_t* wsrc[WU];
#pragma unroll
    for (int j = 0; j < XU; ++j) xrow[j] = min(m0 + j * ROWS_PER_PASS + srow, p.M - 1);
#pragma unroll
    for (int j = 0; j < WU; ++j)
        wsrc[j] = p.w + static_cast<size_t>(min(n0 + j * ROWS_PER_PASS + srow, p.N - 1)) * p.K + schunk * 8;

    // part < 0: the whole tile; part = 0..3: the share of k-slice `part` (the main loop spreads the
    // next tile's loads over the four slices of the current one: a global_load_lds costs the
    // issuing wave ~100 cycles, eight of them in a row at the top of a k-step left the matrix
    // pipe idle for a third of the step)
    auto stage = [&](int buf, int k0, int part) {
        char* xs = smem + buf * STAGE_BYTES;
        char* ws = xs + XT_BYTES;
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            if (part >= 0 && (j < 3 ? j : 3) != part) continue;
            const half_t* xsrc = x_row_ptr<SPATIAL>(p, xrow[j], k0) + schunk * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)xsrc, (lptr_t)(xs + (j * NTHREADS + wave * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WU; ++j) {
            if (part >= 0 && (j < 3 ? j : 3) != part) continue;
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[j] + k0), (lptr_t)(ws + (j * NTHREADS + wave * 64) * 16), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a tile) for the four 16-wide k slices
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = frow * 128 + (((s * 2 + hi) ^ fsw) << 4);

    const bool wave_active = (n0 + wn * (NT * 32)) < p.N;   // N is a multiple of NT*32 per wave
    const int nk = p.K / BK;

    // direct activation fragments (XDIRECT): xnext = k-step t+1 in flight, xcur = k-step t in use
    half8 xcur[4], xnext[4];
    const int xdrow = min(m0 + wm * 32 + frow, p.M - 1);
    auto load_x_direct = [&](int k0) {
        const half_t* xp = x_row_ptr<SPATIAL>(p, xdrow, k0) + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) xnext[s] = *reinterpret_cast<const half8*>(xp + 16 * s);
    };

    // the first tile goes out before anything else touches the memory pipeline
    stage(0, 0, -1);
    if constexpr (XDIRECT) load_x_direct(0);
    if constexpr (STAGES == 3) {
        if (nk > 1) stage(1, BK, -1);
    }

    // The accumulators start at the bias (arithmetic policy: y = ((bias + p_0) + p_1) + ...), so
    // the epilogue has no bias traffic at all. acc[nt][mt][r] belongs to channel
    // n0 + (wn*NT + nt)*32 + 8*(r>>2) + 4*hi + (r&3).
    float16v acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        float16v init;
#pragma unroll
        for (int r = 0; r < 16; ++r) init[r] = 0.f;
        if (p.bias != nullptr && wave_active) {
            const half_t* bp = p.bias + n0 + (wn * NT + a) * 32 + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const half4 b4 = *reinterpret_cast<const half4*>(bp + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) init[4 * g + e] = static_cast<float>(b4[e]);
            }
        }
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = init;
    }

    // first residual: whole rows, 16 B per lane, consumed after the main loop. The loads are
    // issued inside the main loop (first k-step) so that they are neither in front of the first
    // tile in the memory pipeline nor drained by the first barrier's vmcnt(0).
    half8 rpre[NRES >= 1 ? OUNITS : 1];
    auto load_residual = [&]() {
        if constexpr (NRES >= 1) {
#pragma unroll
            for (int j = 0; j < OUNITS; ++j) {
                const int u = j * NTHREADS + tid;
                const int row = min(m0 + u / OCH, p.M - 1);
                const int ch = min(n0 + (u % OCH) * 8, p.N - 8);
                rpre[j] = *reinterpret_cast<const half8*>(p.r1 + static_cast<size_t>(row) * p.ldr1 + ch);
            }
        }
    };

    const float4* tab = nullptr;
    if constexpr (ACT == ACT_WSILU) {
        // WSiLU coefficient table -> LDS (behind the stages); visible after the first barrier
        float4* t = reinterpret_cast<float4*>(smem + STAGES * STAGE_BYTES);
        for (int i = tid; i < WSILU_SEGMENTS; i += NTHREADS) t[i] = p.wsilu[i];
        tab = t;
    }
    if constexpr (STAGES == 3) load_residual();     // (tuning variant only: waited for with tile 0)
    stamp();                                                       // 1: prologue issued
    for (int t = 0; t < nk; ++t) {
        int cur;
        if constexpr (STAGES == 2) {
            __syncthreads();             // tile t landed (vmcnt(0)) and buffer (t+1)&1 is free
            if (t < 8) stamp();                                    // 2..9: k-step t may start
            if (t == 0) load_residual();
            cur = t & 1;
            if constexpr (XDIRECT) {
#pragma unroll
                for (int s = 0; s < 4; ++s) xcur[s] = xnext[s];      // landed: the barrier drained vmcnt
                if (t + 1 < nk) load_x_direct((t + 1) * BK);
            }
        } else {
            // Two tiles in flight: wait only for the OLDER one (counted vmcnt), keep the younger
            // across the barrier. __syncthreads() would drain the LDS-DMA queue, hence the raw
            // barrier. After the barrier every wave has finished computing tile t-1, so its
            // buffer ((t+2) % 3) may be refilled.
            if (t + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XU + WU) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = t % 3;
            if (t + 2 < nk) stage((t + 2) % 3, (t + 2) * BK, -1);
        }
        const char* xs = smem + cur * STAGE_BYTES + wm * (MT * 32 * 128);
        const char* ws = smem + cur * STAGE_BYTES + XT_BYTES + wn * (NT * 32 * 128);
        if (wave_active) {
            // fragments of k-slice s+1 are fetched while the MFMAs of slice s run (register
            // double buffering; hipcc otherwise waits for all six reads in front of every slice)
            half8 xf[2][MT], wf[2][NT];
            if constexpr (!XDIRECT) {
#pragma unroll
                for (int i = 0; i < MT; ++i) xf[0][i] = *reinterpret_cast<const half8*>(xs + i * (32 * 128) + foff[0]);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) wf[0][i] = *reinterpret_cast<const half8*>(ws + i * (32 * 128) + foff[0]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) {
                    if constexpr (!XDIRECT) {
#pragma unroll
                        for (int i = 0; i < MT; ++i)
                            xf[(s + 1) & 1][i] = *reinterpret_cast<const half8*>(xs + i * (32 * 128) + foff[s + 1]);
                    }
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        wf[(s + 1) & 1][i] = *reinterpret_cast<const half8*>(ws + i * (32 * 128) + foff[s + 1]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const half8 xfrag = XDIRECT ? xcur[s] : xf[s & 1][mt];
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s & 1][nt], xfrag, acc[nt][mt], 0, 0, 0);
                    }
                if constexpr (STAGES == 2) {
                    if (t + 1 < nk) stage((t + 1) & 1, (t + 1) * BK, s);
                }
            }
        } else if constexpr (STAGES == 2) {
            if (t + 1 < nk) stage((t + 1) & 1, (t + 1) * BK, -1);
        }
    }

    // ---- epilogue. acc[nt][mt][r]: pixel m = m0 + (wm*MT + mt)*32 + (lane&31),
    //      channel n = n0 + (wn*NT + nt)*32 + 8*(r>>2) + 4*hi + (r&3)
    // The block's output tile goes through LDS (the stage buffers are dead now) so that global
    // memory sees whole 128-B lines: per-lane stores at a pixel stride touch a different line per
    // lane and were 2/3 of the kernel time. The first residual arrives the same way (prefetched
    // row-wise into registers at kernel start, parked in the LDS tile here).
    __syncthreads();                                   // every wave is done with the stage buffers
    stamp_no = 10;
    stamp();                                                       // 10: main loop done
    char* otile = smem;
    constexpr int SWZ = OCH >= 8 ? 7 : OCH - 1;        // rows narrower than 128 B swizzle inside the row
    auto oaddr = [&](int row, int cidx) {              // 16-B chunk cidx of tile row `row`, bank swizzled
        return otile + row * (BNO * 2) + (((cidx & ~SWZ) | ((cidx & SWZ) ^ (row & SWZ))) << 4);
    };
    // WSiLU table gathers are the epilogue's LDS hot spot (128 random 16-byte reads per lane and
    // tile; measured 39 % of all LDS cycles lost to bank conflicts on the plain table). The dead
    // stage area behind the output tile takes R interleaved copies so that lane l reads copy
    // l & (R-1): conflict-free for R = 16, at most 16/R lanes per slot otherwise.
    constexpr int OT_BYTES = BM * BNO * 2;
    constexpr int AREA = STAGES * STAGE_BYTES;
    constexpr int R = (ACT != ACT_WSILU) ? 1
                    : (OT_BYTES + 16 * WSILU_TABLE_BYTES <= AREA) ? 16
                    : (OT_BYTES + 8 * WSILU_TABLE_BYTES <= AREA) ? 8
                    : (OT_BYTES + 4 * WSILU_TABLE_BYTES <= AREA) ? 4
                    : (OT_BYTES + 2 * WSILU_TABLE_BYTES <= AREA) ? 2 : 1;
    if constexpr (R > 1) {
        float4* rep = reinterpret_cast<float4*>(smem + AREA - R * WSILU_TABLE_BYTES);
        const float4* base = reinterpret_cast<const float4*>(smem + AREA);
        for (int i = tid; i < R * WSILU_SEGMENTS; i += NTHREADS) rep[i] = base[i / R];
        __syncthreads();
        tab = rep + (lane & (R - 1));
    }
    if constexpr (NRES >= 1) {
#pragma unroll
        for (int j = 0; j < OUNITS; ++j) {
            const int u = j * NTHREADS + tid;
            *reinterpret_cast<half8*>(oaddr(u / OCH, u % OCH)) = rpre[j];
        }
        __syncthreads();
    }
    if (wave_active) {
        const int ntile = wn * (NT * 32);              // channel offset of this wave inside the tile
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = (wm * MT + mt) * 32 + (lane & 31);
            const int m = m0 + row;
            if constexpr (CHUNK) {
#pragma unroll
                for (int np = 0; np < NT / 2; ++np) {
                    // s = chunk sum of wsilu(acc) over 4 adjacent channels (arith.h: one fma chain per group)
                    float s[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int nt = 2 * np + h;
                        if constexpr (ACT == ACT_WSILU) {
                            wsilu_chunk16<R>(acc[nt][mt], s[h], tab);
                        } else {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                s[h][g] = ((acc[nt][mt][4 * g] + acc[nt][mt][4 * g + 1]) + acc[nt][mt][4 * g + 2]) + acc[nt][mt][4 * g + 3];
                        }
                    }
                    // lower half-wave collects the 8 outputs of tile 2np, upper half-wave those of 2np+1
                    half8 o;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s[0][g]),
                                                                         __float_as_uint(s[1][g]), false, false);
                        o[2 * g] = to_half(__uint_as_float(sw[0]));
                        o[2 * g + 1] = to_half(__uint_as_float(sw[1]));
                    }
                    *reinterpret_cast<half8*>(oaddr(row, ((ntile + np * 64) >> 5) + hi)) = o;
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {          // pair of 4-channel groups (2pr, 2pr+1)
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(
                                __float_as_uint(acc[nt][mt][8 * pr + e]),
                                __float_as_uint(acc[nt][mt][8 * pr + 4 + e]), false, false);
                            v[e] = __uint_as_float(sw[0]);
                            v[4 + e] = __uint_as_float(sw[1]);
                        }
                        // this lane now holds channels cb .. cb+7 of pixel m
                        const int ct = ntile + nt * 32 + 16 * pr + 8 * hi;     // inside the tile
                        const int cb = n0 + ct;
                        if constexpr (ACT == ACT_WSILU) wsilu8<R>(v, tab);
                        half8* slot = reinterpret_cast<half8*>(oaddr(row, ct >> 3));
                        if constexpr (NRES >= 1) {
                            const half8 r8 = *slot;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(r8[e]);
                        }
                        if constexpr (NRES >= 2) {
                            const half8 r8 = *reinterpret_cast<const half8*>(
                                p.r2 + static_cast<size_t>(min(m, p.M - 1)) * p.ldr2 + cb);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(r8[e]);
                        }
                        if constexpr (QUANT) {
                            const half8 q8 = *reinterpret_cast<const half8*>(p.q + cb);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = v[e] * static_cast<float>(q8[e]);
                        }
                        half8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
                        if (p.q2 != nullptr) {
                            const half8 q8 = *reinterpret_cast<const half8*>(p.q2 + cb);
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = hmul(o[e], q8[e]);
                        }
                        *slot = o;
                    }
            }
        }
    }
    __syncthreads();
    stamp();                                                       // 11: epilogue math done, tile in LDS
    // ---- whole-line stores: consecutive lanes write consecutive 16-B chunks of one pixel row
    const int quad = UPSAMPLE ? n0 / p.up_cout : 0;          // transposed conv: which of the four output pixels
    const int n0o = CHUNK ? (n0 >> 2) : (UPSAMPLE ? n0 - quad * p.up_cout : n0);
    const int nout = CHUNK ? (p.N >> 2) : (UPSAMPLE ? p.up_cout : p.N);
#pragma unroll
    for (int j = 0; j < OUNITS; ++j) {
        const int u = j * NTHREADS + tid;
        const int row = u / OCH, ch = u % OCH;
        const int m = m0 + row;
        if (m < p.M && n0o + ch * 8 < nout) {
            size_t orow;
            if constexpr (UPSAMPLE) {
                const int yy = m / p.in_w, xx = m - yy * p.in_w;
                orow = static_cast<size_t>(2 * yy + (quad >> 1)) * (2 * p.in_w) + (2 * xx + (quad & 1));
            } else {
                orow = static_cast<size_t>(m);
            }
            store_line(p.y + orow * p.ldy + n0o + ch * 8, *reinterpret_cast<const half8*>(oaddr(row, ch)));
        }
    }
    stamp();                                                       // 12: stores issued
}

long long* g_timeline = nullptr;       // tools/gemm_timeline.py: device buffer for the in-kernel stamps

}  // namespace

// ---- WSiLU table in device memory (uploaded once, outside any capture); shared with ffn_fused.hip
const float4* wsilu_table_device()
{
    static float4* dev = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        hip_check(hipMalloc(&dev, WSILU_TABLE_BYTES), "hipMalloc(wsilu table)");
        hip_check(hipMemcpy(dev, kWsiluTable, WSILU_TABLE_BYTES, hipMemcpyHostToDevice), "upload wsilu table");
    });
    return dev;
}

namespace {

// ---- optional per-launch timing (bench.py's roofline leg): hipExtLaunchKernel stamps the
// kernel's own begin / end into the two events
struct GemmProfile {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<GemmLaunchInfo> info;
    size_t used = 0;
    std::mutex mu;          // launches may come from several host threads (lanes)
};

GemmProfile& profile()
{
    static GemmProfile g;
    return g;
}

template <int WM, int WN, int MT, int NT, int STAGES, bool SPATIAL, int ACT, bool CHUNK, int NRES, bool QUANT, bool UPSAMPLE,
          bool XDIRECT = false>
void launch_cfg(const ConvGemmParams& p, hipStream_t stream)
{
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, NTHREADS = WM * WN * 64;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    auto kern = conv_gemm_kernel<WM, WN, MT, NT, STAGES, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE, XDIRECT>;
    static std::once_flag attr_once;      // lanes launch from several host threads
    const int smem_bytes = STAGES * ((XDIRECT ? 0 : BM) + BN) * BK * 2 + (ACT == ACT_WSILU ? WSILU_TABLE_BYTES : 0);
    std::call_once(attr_once, [&] {
        hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes),
                  "hipFuncSetAttribute(conv_gemm)");
    });
    GemmProfile& pf = profile();
    if (pf.on) {
        std::lock_guard<std::mutex> lk(pf.mu);
        if (pf.used == pf.events.size()) {
            hipEvent_t a, b;
            hip_check(hipEventCreate(&a), "hipEventCreate");
            hip_check(hipEventCreate(&b), "hipEventCreate");
            pf.events.emplace_back(a, b);
            pf.info.push_back(GemmLaunchInfo{});
        }
        pf.info[pf.used] = GemmLaunchInfo{ p.M, p.N, p.K,
                                           (SPATIAL ? 1 : 0) | (ACT << 1) | (CHUNK ? 4 : 0) | (NRES << 3) |
                                               (QUANT ? 32 : 0) | (UPSAMPLE ? 64 : 0) | (BM << 8) | (BN << 18), 0.f };      // bits 28..31 = kernel family (ops.h): 0 here
        hipExtLaunchKernelGGL(kern, dim3(tiles), dim3(NTHREADS), smem_bytes, stream, pf.events[pf.used].first,
                              pf.events[pf.used].second, 0, p);
        ++pf.used;
    } else {
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHREADS), smem_bytes, stream, p);
    }
    hip_check(hipGetLastError(), "conv_gemm launch");
}

// Tile shape per launch. The L2 -> LDS path delivers ~56 B/clk per CU, a 128x128x64 tile needs
// 64 B/clk to keep the matrix cores busy, a 256x256 one 31 B/clk: the big P8 layers therefore
// run 256-pixel tiles with 8 waves (256x256 when N allows, else 256x192, both exactly one or
// three rounds of the 256 CUs at 1080p), small grids fall back to 128x128 / 64x128 so that the
// chip is still filled.
template <bool SPATIAL, int ACT, bool CHUNK, int NRES, bool QUANT, bool UPSAMPLE>
void launch(ConvGemmParams p, hipStream_t stream)
{
    p.wsilu = (ACT == ACT_WSILU) ? wsilu_table_device() : nullptr;
    p.timeline = g_timeline;
    const long long tiles128 = static_cast<long long>((p.M + 127) / 128) * ((p.N + 127) / 128);
    const long long mt256 = (p.M + 255) / 256;
    static const int force = [] { const char* e = getenv("DCVC_GEMM_CFG"); return e ? atoi(e) : 0; }();
    if (force) {      // tuning experiments only
        switch (force) {
        case 1: launch_cfg<2, 2, 2, 2, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream); return;
        case 2: launch_cfg<2, 2, 2, 2, 3, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream); return;
        case 3: launch_cfg<4, 2, 2, 2, 3, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream); return;
        case 4: launch_cfg<4, 2, 2, 4, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream); return;
        case 5: launch_cfg<2, 2, 1, 2, 3, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream); return;
        case 6: launch_cfg<2, 2, 1, 2, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream); return;
        case 7: launch_cfg<4, 2, 2, 2, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream); return;
        case 8: launch_cfg<2, 2, 2, 4, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream); return;
        case 9: launch_cfg<4, 1, 1, 4, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream); return;
        case 10: launch_cfg<4, 1, 1, 8, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE, true>(p, stream); return;
        case 11:
            if constexpr (!CHUNK) { launch_cfg<4, 1, 1, 6, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE, true>(p, stream); return; }
            break;
        default: break;
        }
    }
    // (transposed conv: a channel tile must not straddle two output pixels - its width divides up_cout)
    const int nq = UPSAMPLE ? p.up_cout : p.N;
    if (nq % 256 == 0 && mt256 * (p.N / 256) >= 224) {
        launch_cfg<4, 2, 2, 4, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream);
        return;
    }
    if constexpr (!CHUNK) {
        if (nq % 192 == 0 && mt256 * (p.N / 192) >= 224) {
            launch_cfg<4, 2, 2, 3, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream);
            return;
        }
    }
    if constexpr (!CHUNK) {
        // narrow layers on small grids (the 3x3 stride-2 conv behind LD's encoder: 8160 x 128 outputs, K = 2304) leave half the
        // chip idle with 64 x 128 tiles: 64 x 64 ones (round 6)
        const long long tiles64x128 = static_cast<long long>((p.M + 63) / 64) * ((p.N + 127) / 128);
        if (tiles64x128 <= 160 && p.K >= 512 && (!UPSAMPLE || p.up_cout % 64 == 0)) {
            launch_cfg<2, 2, 1, 1, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream);
            return;
        }
    }
    if (tiles128 < 640) {
        launch_cfg<2, 2, 1, 2, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream);
    } else {
        launch_cfg<2, 2, 2, 2, 2, SPATIAL, ACT, CHUNK, NRES, QUANT, UPSAMPLE>(p, stream);
    }
}

void check_common(const ConvGemmParams& p)
{
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) {
        throw std::invalid_argument("conv_gemm: empty problem");
    }
    if (p.K % BK != 0 || p.N % 64 != 0) {
        throw std::invalid_argument("conv_gemm: K must be a multiple of 64 and N of 64 (K=" +
                                    std::to_string(p.K) + ", N=" + std::to_string(p.N) + ")");
    }
    if ((p.ldx % 8) || (p.ldy % 8) || (p.r1 && p.ldr1 % 8) || (p.r2 && p.ldr2 % 8)) {
        throw std::invalid_argument("conv_gemm: leading dimensions must be multiples of 8 channels");
    }
}

}  // namespace

void kernels_init()
{
    (void)wsilu_table_device();
    symbols_init();
}

void gemm_timeline_buffer(long long* device_buffer)
{
    g_timeline = device_buffer;
}

void gemm_profile_enable(bool on)
{
    profile().on = on;
}

bool gemm_profile_slot(const GemmLaunchInfo& info, hipEvent_t* start, hipEvent_t* stop)
{
    GemmProfile& pf = profile();
    if (!pf.on) return false;
    std::lock_guard<std::mutex> lk(pf.mu);
    if (pf.used == pf.events.size()) {
        hipEvent_t a, b;
        hip_check(hipEventCreate(&a), "hipEventCreate");
        hip_check(hipEventCreate(&b), "hipEventCreate");
        pf.events.emplace_back(a, b);
        pf.info.push_back(GemmLaunchInfo{});
    }
    pf.info[pf.used] = info;
    *start = pf.events[pf.used].first;
    *stop = pf.events[pf.used].second;
    ++pf.used;
    return true;
}

void gemm_profile_reset()
{
    profile().used = 0;
}

size_t gemm_profile_launches(GemmLaunchInfo* out, size_t cap)
{
    GemmProfile& pf = profile();
    for (size_t i = 0; i < pf.used && i < cap; ++i) {
        float e = 0.f;
        hip_check(hipEventSynchronize(pf.events[i].second), "hipEventSynchronize");
        hip_check(hipEventElapsedTime(&e, pf.events[i].first, pf.events[i].second), "hipEventElapsedTime");
        out[i] = pf.info[i];
        out[i].ms = e;
    }
    return pf.used;
}

void gemm_profile_collect(double* ms, double* flops, long long* launches)
{
    GemmProfile& pf = profile();
    std::vector<GemmLaunchInfo> rec(pf.used);
    gemm_profile_launches(rec.data(), rec.size());
    double t = 0, f = 0;
    for (const GemmLaunchInfo& r : rec) {
        t += r.ms;
        f += 2.0 * r.M * r.N * r.K;
    }
    *ms = t;
    *flops = f;
    *launches = static_cast<long long>(pf.used);
}

// ------------------------------------------------------------------------------------------
// public launchers (ops.h)
// ------------------------------------------------------------------------------------------
void conv1x1(const Conv1x1Desc& d, hipStream_t stream)
{
    ConvGemmParams p{};
    p.x = d.x;  p.ldx = d.ldx;
    p.w = d.w;  p.bias = d.bias;
    p.r1 = d.r1; p.ldr1 = d.ldr1;
    p.r2 = d.r2; p.ldr2 = d.ldr2;
    p.q = d.q;  p.q2 = d.q2;
    p.zeros = nullptr;
    p.y = d.y;  p.ldy = d.ldy;
    p.M = d.pixels; p.N = d.cout; p.K = d.cin;
    check_common(p);
    const int nres = (d.r1 ? 1 : 0) + (d.r2 ? 1 : 0);
    if (d.r2 && !d.r1) {
        throw std::invalid_argument("conv1x1: r2 without r1");
    }
    if (d.chunk_add) {
        if (!d.bias || nres || d.q || d.q2 || d.cout % 256 != 0) {
            throw std::invalid_argument("conv1x1: chunk-add needs bias, no residual/quant, N % 256 == 0");
        }
        if (d.wsilu) launch<false, ACT_WSILU, true, 0, false, false>(p, stream);
        else         launch<false, ACT_NONE, true, 0, false, false>(p, stream);
        return;
    }
    if (d.wsilu) {
        if (nres || d.q) throw std::invalid_argument("conv1x1: wsilu with residual/quant is not a reference op");
        launch<false, ACT_WSILU, false, 0, false, false>(p, stream);
        return;
    }
    if (d.q) {
        if (nres == 0)      launch<false, ACT_NONE, false, 0, true, false>(p, stream);
        else if (nres == 1) launch<false, ACT_NONE, false, 1, true, false>(p, stream);
        else throw std::invalid_argument("conv1x1: quant with two residuals is not a reference op");
        return;
    }
    if (nres == 0)      launch<false, ACT_NONE, false, 0, false, false>(p, stream);
    else if (nres == 1) launch<false, ACT_NONE, false, 1, false, false>(p, stream);
    else                launch<false, ACT_NONE, false, 2, false, false>(p, stream);
}

void conv_kxk(const ConvKxKDesc& d, hipStream_t stream)
{
    ConvGemmParams p{};
    p.x = d.x;  p.ldx = d.ldx;
    p.w = d.w;  p.bias = d.bias;
    p.zeros = d.zeros;
    p.y = d.y;  p.ldy = d.ldy;
    p.in_h = d.in_h; p.in_w = d.in_w;
    p.out_h = (d.in_h + 2 * d.pad - d.ksize) / d.stride + 1;
    p.out_w = (d.in_w + 2 * d.pad - d.ksize) / d.stride + 1;
    p.cin = d.cin; p.ksize = d.ksize; p.stride = d.stride; p.pad = d.pad;
    p.M = p.out_h * p.out_w; p.N = d.cout; p.K = d.ksize * d.ksize * d.cin;
    check_common(p);
    if (d.cin % BK != 0 || !d.zeros) {
        throw std::invalid_argument("conv_kxk: Cin must be a multiple of 64 and a zero page is required");
    }
    launch<true, ACT_NONE, false, 0, false, false>(p, stream);
}

void tconv2x2(const TConv2x2Desc& d, hipStream_t stream)
{
    // out[2y+dy][2x+dx][co] = sum_ci x[y][x][ci] * w[(dy*2+dx)][co][ci]   (no bias: the reference
    // folds SubpelConv2x(kernel 1) into a stride-2 transposed conv, layers_proxy.cpp:320-323)
    // ONE launch (round 6; four until then): the four weight matrices are one [4 * cout][cin] matrix, a channel tile of the
    // product belongs to one of the four output pixels and is scattered there by the epilogue's whole-line stores.
    if (d.cout % 128 != 0) throw std::invalid_argument("tconv2x2: cout must be a multiple of 128 (one channel tile per output pixel)");
    ConvGemmParams p{};
    p.x = d.x;  p.ldx = d.ldx;
    p.w = d.w;
    p.bias = nullptr;
    p.y = d.y;  p.ldy = d.ldy;
    p.in_h = d.in_h; p.in_w = d.in_w;
    p.up_cout = d.cout;
    p.M = d.in_h * d.in_w; p.N = 4 * d.cout; p.K = d.cin;
    check_common(p);
    launch<false, ACT_NONE, false, 0, false, true>(p, stream);
}

}  // namespace dcvc
