// ffn_fused.hip - the FFN half of a DepthConvBlock in ONE launch:
//
//     t = chunk_add(WSiLU(W0 * y1 + b0))          ffn.0   (conv1x1_bias_wsilu_chunk_add)
//     out = W2 * t + b2 + y1 [+ x] [* q]          ffn.2   (conv1x1_bias_shortcut[2][_with_quant])
//
// Reference: layers_proxy.cpp:84-98 runs these as two CUTLASS launches with the 4x-expanded tensor
// fused away and `t` ([pixels][C_ffn] fp16) round-tripping through memory. Here `t` never leaves
// the CU: a workgroup owns 128 pixel rows, walks over ffn.0's output channels in tiles of 256
// (= 64 channels of t after the chunk-add), and after each tile feeds the 64 fresh channels of t
// straight from LDS into ffn.2's accumulators, which stay in registers for the whole walk:
//
//     for j in 0 .. C_ffn/64 - 1:
//         acc0[128 x 256]  = b0 + W0[256j .. 256j+255] . y1          (K = C, streamed in 64-wide slabs)
//         T[128 x 64]      = chunk_add(WSiLU(acc0))  -> fp16 -> LDS   (the fp16 rounding point of t)
//         acc2[128 x C]   += W2[:, 64j .. 64j+63] . T                (K = 64)
//     out = acc2 + y1 [+ x] [* q] -> fp16
//
// ffn.2's contraction index runs over the channels of t in ascending order, 16 at a time, into one
// accumulator per output - exactly the order of the two-launch path and of the oracle, so the
// result is bit-identical to it (arithmetic policy: DESIGN.md section 2). What the fusion removes:
// the write + read of t (2 x 25 MB per block at 1080p), ffn.2's re-read of y1 as a GEMM operand is
// gone as well (T comes from LDS), one kernel boundary, and ffn.2's prologue / epilogue ramps.
//
// Geometry: 512 threads = 8 waves as 2 (pixel halves of 64) x 4 (channel quarters); LDS = two
// stages of (y1 slab 128x64 + W0 slab 256x64) = 96 KB + the 4 KB WSiLU table; between two channel
// tiles the dead stage area holds T (16 KB), the W2 slab [C][64] and the replicated WSiLU table.
#include "arith.h"
#include <mutex>
#include "ops.h"
#include "wsilu_table.h"

#include <hip/hip_ext.h>

#include <cstdlib>

namespace dcvc {

const float4* wsilu_table_device();      // conv_gemm.hip

namespace {

constexpr int BK = 64;
constexpr int NTHREADS = 512;
constexpr int BM = 128;                  // pixel rows per workgroup
constexpr int BNA = 256;                 // ffn.0 output channels per tile (64 channels of t)
constexpr int XT_BYTES = BM * BK * 2;    // 16 KB
constexpr int WT_BYTES = BNA * BK * 2;   // 32 KB
constexpr int STAGE_BYTES = XT_BYTES + WT_BYTES;
constexpr int AREA = 2 * STAGE_BYTES;    // 96 KB
constexpr int TABLE_BYTES = WSILU_SEGMENTS * 16;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct FfnParams {
    const half_t* x;      // y1 [M][ldx], first C channels: input of ffn.0 and first residual of ffn.2
    const half_t* w0;     // [4*CF][C]
    const half_t* b0;     // [4*CF]
    const half_t* w2;     // [C][CF]
    const half_t* b2;     // [C]
    const half_t* r2;     // optional second residual [M][ldr2]
    const half_t* q;      // optional fused per-channel scale [C]
    const half_t* q2;     // optional scale applied to the rounded output [C]
    const float4* wsilu;
    half_t* y;            // [M][ldy]; may alias x (row-local read-before-write)
    int ldx, ldr2, ldy;
    int M, C, CF;
};

// NT2 = C / 128: 32-channel output tiles per wave in ffn.2
template <int NT2, bool RES2, bool QUANT>
__global__ void __launch_bounds__(NTHREADS)
ffn_fused_kernel(const FfnParams p)
{
    constexpr int C = NT2 * 128;
    constexpr int XU = BM * 8 / NTHREADS;            // 2: 16-B units per thread and y1 slab
    constexpr int WU = BNA * 8 / NTHREADS;           // 4: ... and W0 slab
    constexpr int W2U = C * 8 / NTHREADS;            // W2 slab [C][64]
    constexpr int T_BYTES = BM * 128;                // T tile: 128 rows x 64 ch fp16
    // the W2 slab has its own buffer behind the table (fetched while the main loop runs) when LDS
    // allows; otherwise it shares the dead stage area and its latency is exposed
    // (measured: a dedicated buffer fetched at the top of the tile is no faster for C <= 256 and slower
    // for C = 384 - the slab then competes with the first stage - so the shared placement is used)
    constexpr bool SLAB_OWN = false;
    constexpr int SLAB2_OFF = SLAB_OWN ? AREA + TABLE_BYTES : T_BYTES;
    constexpr int REP_OFF = SLAB_OWN ? T_BYTES : T_BYTES + C * 128;
    constexpr int REP_FREE = AREA - REP_OFF;
    constexpr int R = REP_FREE >= 16 * TABLE_BYTES ? 16 : REP_FREE >= 8 * TABLE_BYTES ? 8
                    : REP_FREE >= 4 * TABLE_BYTES ? 4 : REP_FREE >= 2 * TABLE_BYTES ? 2 : 1;
    constexpr int OCH = C / 8;                       // 16-B chunks per output row
    constexpr int OUNITS = BM * OCH / NTHREADS;
    static_assert(REP_OFF <= AREA && BM * C * 2 <= AREA, "W2 slab / output tile must fit the stage area");
    static_assert(W2U * NTHREADS * 16 == C * 128, "whole 16-B units");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int hi = lane >> 5;
    const int frow = lane & 31;
    const int m0 = blockIdx.x * BM;

    // ---- staging plan (as conv_gemm.hip): unit u -> row u>>3, physical chunk u&7,
    //      logical chunk = physical ^ ((row>>1)&7)
    const int srow = tid >> 3;                        // 0..63
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
    const half_t* xsrc[XU];
#pragma unroll
    for (int j = 0; j < XU; ++j)
        xsrc[j] = p.x + static_cast<size_t>(min(m0 + j * 64 + srow, p.M - 1)) * p.ldx + schunk * 8;

    auto stage = [&](int buf, int jn, int k0, int part) {
        char* xs = smem + buf * STAGE_BYTES;
        char* ws = xs + XT_BYTES;
#pragma unroll
        for (int j = 0; j < XU; ++j) {
            if (part >= 0 && j != part) continue;
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[j] + k0), (lptr_t)(xs + (j * NTHREADS + wave * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < WU; ++j) {
            if (part >= 0 && j != part) continue;
            const half_t* wsrc = p.w0 + static_cast<size_t>(jn * BNA + j * 64 + srow) * C + k0 + schunk * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)wsrc, (lptr_t)(ws + (j * NTHREADS + wave * 64) * 16), 16, 0, 0);
        }
    };

    const int fsw = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = frow * 128 + (((s * 2 + hi) ^ fsw) << 4);

    constexpr int nk = C / BK;
    const int nj = p.CF / 64;

    // ---- ffn.2 accumulators, alive over the whole walk, start at b2:
    //      acc2[nt][mt][r] = channel (wn*NT2 + nt)*32 + 8*(r>>2) + 4*hi + (r&3), pixel (wm*2 + mt)*32 + frow
    float16v acc2[NT2][2];
#pragma unroll
    for (int a = 0; a < NT2; ++a) {
        float16v init;
        const half_t* bp = p.b2 + (wn * NT2 + a) * 32 + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const half4 b4 = *reinterpret_cast<const half4*>(bp + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) init[4 * g + e] = static_cast<float>(b4[e]);
        }
        acc2[a][0] = init;
        acc2[a][1] = init;
    }

    const float4* base_tab = reinterpret_cast<const float4*>(smem + AREA);
    {
        float4* t = reinterpret_cast<float4*>(smem + AREA);
        for (int i = tid; i < WSILU_SEGMENTS; i += NTHREADS) t[i] = p.wsilu[i];
    }

    auto load_w2_slab = [&](int jn) {        // [C][64] of this t-tile, stage layout (128-B rows, swizzled source)
#pragma unroll
        for (int j = 0; j < W2U; ++j) {
            const half_t* src = p.w2 + static_cast<size_t>(j * 64 + srow) * p.CF + jn * 64 + schunk * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + SLAB2_OFF + (j * NTHREADS + wave * 64) * 16), 16, 0, 0);
        }
    };

    for (int jn = 0; jn < nj; ++jn) {
        stage(0, jn, 0, -1);
        if constexpr (SLAB_OWN) load_w2_slab(jn);
        // ---- ffn.0 accumulators of this channel tile, start at b0
        float16v acc0[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float16v init;
            const half_t* bp = p.b0 + jn * BNA + (wn * 2 + a) * 32 + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const half4 b4 = *reinterpret_cast<const half4*>(bp + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) init[4 * g + e] = static_cast<float>(b4[e]);
            }
            acc0[a][0] = init;
            acc0[a][1] = init;
        }
        for (int t = 0; t < nk; ++t) {
            __syncthreads();                 // slab t landed (vmcnt(0)), buffer (t+1)&1 is free
            const int cur = t & 1;
            const char* xs = smem + cur * STAGE_BYTES + wm * (2 * 32 * 128);
            const char* ws = smem + cur * STAGE_BYTES + XT_BYTES + wn * (2 * 32 * 128);
            half8 xf[2][2], wf[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xf[0][i] = *reinterpret_cast<const half8*>(xs + i * (32 * 128) + foff[0]);
                wf[0][i] = *reinterpret_cast<const half8*>(ws + i * (32 * 128) + foff[0]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        xf[(s + 1) & 1][i] = *reinterpret_cast<const half8*>(xs + i * (32 * 128) + foff[s + 1]);
                        wf[(s + 1) & 1][i] = *reinterpret_cast<const half8*>(ws + i * (32 * 128) + foff[s + 1]);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc0[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s & 1][nt], xf[s & 1][mt], acc0[nt][mt], 0, 0, 0);
                if (t + 1 < nk) stage((t + 1) & 1, jn, (t + 1) * BK, s);
            }
        }
        __syncthreads();                     // every wave is done with the stage buffers
        if constexpr (!SLAB_OWN) load_w2_slab(jn);
        // ---- replicated WSiLU table (conflict-free gathers, see conv_gemm.hip)
        const float4* tab = base_tab;
        if constexpr (R > 1) {
            float4* rep = reinterpret_cast<float4*>(smem + REP_OFF);
            for (int i = tid; i < R * WSILU_SEGMENTS; i += NTHREADS) rep[i] = base_tab[i / R];
            __syncthreads();
            tab = rep + (lane & (R - 1));
        }
        // ---- T = chunk_add(WSiLU(acc0)) -> fp16 -> LDS. T row = pixel, 8 chunks of 8 channels;
        //      chunk c of row r lives at r*128 + ((c ^ (r & 7)) << 4)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = (wm * 2 + mt) * 32 + frow;
            float sum[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                wsilu_chunk16<R>(acc0[h][mt], sum[h], tab);
            }
            half8 o;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum[0][g]), __float_as_uint(sum[1][g]), false, false);
                o[2 * g] = to_half(__uint_as_float(sw[0]));
                o[2 * g + 1] = to_half(__uint_as_float(sw[1]));
            }
            const int chunk = wn * 2 + hi;          // channels (wn*2 + hi)*8 .. +7 of this t-tile
            *reinterpret_cast<half8*>(smem + row * 128 + ((chunk ^ (row & 7)) << 4)) = o;
        }
        __syncthreads();                     // T complete, W2 slab landed (vmcnt(0))
        // ---- acc2 += W2[:, 64 jn ..] . T   (K = 64: four 16-wide slices, ascending)
        {
            const char* ws2 = smem + SLAB2_OFF + wn * (NT2 * 32 * 128);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                half8 tf[2], w2f[NT2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int row = (wm * 2 + mt) * 32 + frow;
                    tf[mt] = *reinterpret_cast<const half8*>(smem + row * 128 + (((s * 2 + hi) ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) w2f[nt] = *reinterpret_cast<const half8*>(ws2 + nt * (32 * 128) + foff[s]);
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc2[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[nt], tf[mt], acc2[nt][mt], 0, 0, 0);
            }
        }
        __syncthreads();                     // T / slab are free again
    }

    // ---- final epilogue of ffn.2: out = acc2 + y1 [+ r2] [* q] -> fp16 [* q2], through LDS so
    //      that memory sees whole lines. Row r, chunk c lives at r*C*2 + (swizzled c) * 16.
    auto oaddr = [&](int row, int cidx) {
        return smem + row * (C * 2) + (((cidx & ~7) | ((cidx & 7) ^ (row & 7))) << 4);
    };
#pragma unroll
    for (int j = 0; j < OUNITS; ++j) {           // first residual = the FFN input rows
        const int u = j * NTHREADS + tid;
        const int row = u / OCH, ch = u % OCH;
        const int m = min(m0 + row, p.M - 1);
        *reinterpret_cast<half8*>(oaddr(row, ch)) = *reinterpret_cast<const half8*>(p.x + static_cast<size_t>(m) * p.ldx + ch * 8);
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = (wm * 2 + mt) * 32 + frow;
        const int m = min(m0 + row, p.M - 1);
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2[nt][mt][8 * pr + e]),
                                                                     __float_as_uint(acc2[nt][mt][8 * pr + 4 + e]), false, false);
                    v[e] = __uint_as_float(sw[0]);
                    v[4 + e] = __uint_as_float(sw[1]);
                }
                const int cb = (wn * NT2 + nt) * 32 + 16 * pr + 8 * hi;
                half8* slot = reinterpret_cast<half8*>(oaddr(row, cb >> 3));
                const half8 r8 = *slot;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(r8[e]);
                if constexpr (RES2) {
                    const half8 s8 = *reinterpret_cast<const half8*>(p.r2 + static_cast<size_t>(m) * p.ldr2 + cb);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(s8[e]);
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
    __syncthreads();
#pragma unroll
    for (int j = 0; j < OUNITS; ++j) {
        const int u = j * NTHREADS + tid;
        const int row = u / OCH, ch = u % OCH;
        const int m = m0 + row;
        if (m < p.M) {
            *reinterpret_cast<half8*>(p.y + static_cast<size_t>(m) * p.ldy + ch * 8) = *reinterpret_cast<const half8*>(oaddr(row, ch));
        }
    }
}

template <int NT2, bool RES2, bool QUANT>
void launch(const FfnParams& p, hipStream_t stream)
{
    auto kern = ffn_fused_kernel<NT2, RES2, QUANT>;
    static std::once_flag attr_once;      // lanes launch from several host threads
    constexpr int smem_bytes = AREA + TABLE_BYTES;
    std::call_once(attr_once, [&] {
        hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes),
                  "hipFuncSetAttribute(ffn_fused)");
    });
    hipEvent_t ev0, ev1;      // bench.py's roofline pass: ffn.0 + ffn.2 as one record (family 3 in bits 28..30, ops.h)
    if (gemm_profile_slot(GemmLaunchInfo{p.M, p.C, 5 * p.CF, 0x30000000, 0.f}, &ev0, &ev1)) {
        hipExtLaunchKernelGGL(kern, dim3((p.M + BM - 1) / BM), dim3(NTHREADS), smem_bytes, stream, ev0, ev1, 0, p);
    } else {
        hipLaunchKernelGGL(kern, dim3((p.M + BM - 1) / BM), dim3(NTHREADS), smem_bytes, stream, p);
    }
    hip_check(hipGetLastError(), "ffn_fused launch");
}

template <int NT2>
void launch_variant(const FfnParams& p, hipStream_t stream)
{
    const bool res2 = p.r2 != nullptr, quant = p.q != nullptr;
    if (res2 && quant) launch<NT2, true, true>(p, stream);
    else if (res2) launch<NT2, true, false>(p, stream);
    else if (quant) launch<NT2, false, true>(p, stream);
    else launch<NT2, false, false>(p, stream);
}

}  // namespace

bool ffn_fused_supported(int pixels, int c, int cffn)
{
    // DCVC_FFN_FUSED: 0 = never (A/B measurements), 2 = whenever the shape allows (parity tests on
    // small pictures), unset / 1 = when it pays
    static const int mode = [] { const char* e = getenv("DCVC_FFN_FUSED"); return e != nullptr ? atoi(e) : 1; }();
    if (mode == 0) return false;
    const bool shape_ok = (c == 128 || c == 256 || c == 384) && cffn % 64 == 0 && cffn >= 64 && pixels > 0;
    // one 128-row strip per workgroup: worth it once the strips fill the chip. Measured on MI355X
    // (1080p): C = 128 / 256 (the inter models' half-width blocks) 11.0 vs 17.7 us and 25.2 vs 30.0 us,
    // LD end to end 157 vs 138 pictures/s; C = 384 (intra) is a wash (83 vs 85 us, and slower end to
    // end) because the per-tile WSiLU epilogue and load latencies are exposed six times per strip.
    return shape_ok && (mode == 2 || (pixels >= 128 * 192 && c <= 256));
}

void ffn_fused(const FfnFusedDesc& d, hipStream_t stream)
{
    if (!(d.c == 128 || d.c == 256 || d.c == 384) || d.cffn % 64 != 0 || d.cffn < 64 || d.pixels <= 0) {
        throw std::invalid_argument("ffn_fused: unsupported shape (C in {128, 256, 384}, C_ffn a multiple of 64)");
    }
    if (d.ldx % 8 != 0 || d.ldy % 8 != 0 || (d.r2 != nullptr && d.ldr2 % 8 != 0)) {
        throw std::invalid_argument("ffn_fused: leading dimensions must be multiples of 8");
    }
    FfnParams p{};
    p.x = d.x; p.ldx = d.ldx; p.w0 = d.w0; p.b0 = d.b0; p.w2 = d.w2; p.b2 = d.b2;
    p.r2 = d.r2; p.ldr2 = d.ldr2; p.q = d.q; p.q2 = d.q2; p.y = d.y; p.ldy = d.ldy;
    p.M = d.pixels; p.C = d.c; p.CF = d.cffn;
    p.wsilu = wsilu_table_device();
    switch (d.c) {
    case 128: launch_variant<1>(p, stream); break;
    case 256: launch_variant<2>(p, stream); break;
    default: launch_variant<3>(p, stream); break;
    }
}

}  // namespace dcvc
