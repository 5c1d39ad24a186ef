// dcb_nsplit_kernel.h - a DepthConvBlock behind its depthwise conv in ONE launch, "N-split" form:
//
//     y1 = W3 * t2 + b3' + x                          dc.3 (+ folded depthwise bias) + block input
//     t  = chunk_add(WSiLU(W0 * y1 + b0))             ffn.0   (4x expansion, never materialised)
//     y  = (W2 * t + b2 + y1 [+ x]) [* q] -> fp16 [* q2]     ffn.2 (+ block shortcut, + quant scales)
//     [t1' = WSiLU(W1' * y + b1')]                    dc.0 of the NEXT block of a chain (optional)
//
// Reference: DepthConvBlockProxy::forward, layers_proxy.cpp:71-101 (3-4 CUTLASS launches). Same contract, same
// arithmetic (contraction order, bias-initialised accumulators, epilogue order, rounding points) as
// conv_gemm.hip / dcb_core.hip: bit-identical to both (tests/test_kernels_gpu.py) and to the oracle.
//
// Round 3. dcb_core.hip keeps a wave's ACTIVATIONS in registers and streams the weights through LDS, shared by the
// four waves of a workgroup; measured (profiles/r03_core_bench_ablation.txt) that sharing costs more than it saves:
// per 16-MFMA slab a lone in-order wave also issues 4 LDS-DMA pieces (21 % of the kernel), one barrier (12 %), one
// ds_read_b128 per MFMA, and the four waves run in lockstep into the same address path - 188 k cycles where the
// matrix cores need 64 k. Here the roles are swapped:
//
//   * a workgroup owns PX = 32 * PXT pixels (64 at picture resolution / 8), whose activations live in LDS
//     (A = [PX][CI], B = [PX][C] fp16, XOR-swizzled 16-byte chunks: layer input and layer output ping-pong);
//   * a wave owns a QUARTER OF THE OUTPUT CHANNELS of every layer and all PX pixels: its weight fragments
//     come straight from L2 into registers (global_load_dwordx4 of a pre-packed, per-wave linear stream: one
//     contiguous KB per MFMA "A" operand, prefetched 16 fragments = 4 k-slices ahead), every fragment feeds PXT
//     MFMAs, activation ("B") fragments are PXT ds_read_b128 per k-slice for 3-4 * PXT MFMAs;
//   * no barrier inside a layer (4 per block), no LDS-DMA in the main loop, the waves drift apart freely;
//   * workgroups are persistent (one per CU, tiles round-robin): constants once, the next tile's t2 (whole rows by
//     LDS-DMA) and x (registers) requested behind the current tile's last contractions; outputs leave straight from
//     the epilogues' registers (NS_DIRECT).
//   Cost: every workgroup streams the block's weights itself (2 MB per 64 pixels from L2 instead of per 128) -
//   64 B/clk/CU at full matrix-core rate, the L1 fill rate; L2-resident because every CU streams the same bytes.
//
// Template <C, CI, PXT, NEXT>: C = block width, CI = inner width (dc.0 / depthwise output and chunk-added ffn width:
// C for the full-width blocks, C / 2 for the `dcb2` blocks of the inter models, layers.py:128-159), PXT = 32-pixel
// tiles per workgroup. <384, 384, 2> is the intra codec's encoder / decoder block; <512, 512, 1> (32 pixels per workgroup:
// 255 workgroups on the 68 x 120 grid) the prior networks at picture resolution / 16; <512, 256, 2> the
// hierarchical models' feature / encoder / decoder chains, <256, 128, 2> the low-delay model's; <256, 256, 2> the
// hierarchical models' reconstruction heads, <768, 768, 1> their prior fusion at / 16.
// This header holds the kernel and its launcher; dcb_nsplit_<shape>.hip instantiate it (one translation unit per block
// shape: the fully unrolled kernels take minutes to compile, the build runs the units in parallel), dcb_nsplit.hip
// holds the weight packing and the host entry points.
#pragma once
#include "arith.h"
#include "ops.h"
#include "wsilu_table.h"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace dcvc {

const float4* wsilu_table_device();      // conv_gemm.hip

namespace nsplit {

constexpr int NTHREADS = 256;
#ifndef NS_RING
#define NS_RING 16
#endif
constexpr int RING_DEFAULT = NS_RING;    // weight fragments in flight per wave (4 registers each)
// Round-4 switches (each bit-identical; defaults = what measured best, profiles/r04_core_bench_*.txt):
//   NS_XEARLY   where the next tile's x is requested: 0 = in front of dc.0's epilogue (round 3), 1 = in front of the last
//               ffn.0 pass's (exposed) epilogue - one transfer burst per exposed epilogue: x | t2 + y | t1' -, 2 = with t2
//               behind ffn.2's MFMAs (x + t2 + y | t1')
//   NS_WAITLAST within a k-slice the LAST fragment of the slice feeds the first MFMAs: one counted wait per slice
#ifndef NS_XEARLY
#define NS_XEARLY 2
#endif
#ifndef NS_WAITLAST
#define NS_WAITLAST 1
#endif
//   NS_BUFLOAD  weight fragments by buffer_load (resource = the packed stream, lane offset + 12-bit immediate + a scalar
//               offset per 4 KB) instead of global_load with a VALU addition per fragment beyond the immediate's reach
//   NS_ADDR8    the 8 distinct swizzled LDS row addresses of the activation fragments (k-slice mod 8) live in registers,
//               everything else of a fragment's address is the instruction's immediate: no VALU per fragment read
#ifndef NS_BUFLOAD
#define NS_BUFLOAD 1
#endif
#ifndef NS_ADDR8
#define NS_ADDR8 1
#endif
constexpr bool WAITLAST = NS_WAITLAST != 0;
constexpr int XEARLY = NS_XEARLY;
constexpr bool BUFLOAD = NS_BUFLOAD != 0, ADDR8 = NS_ADDR8 != 0;
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
constexpr int R = 4;                     // interleaved copies of the WSiLU table
constexpr int TABLE_BYTES = WSILU_SEGMENTS * 16;
constexpr int align16k(int bytes) { return (bytes + 16383) & ~16383; }

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N). The weight ring below is indexed ONLY through such
// constants: with plain (later unrolled) loop counters its promotion to registers depended on the order of LLVM's
// unroll / SROA passes and came and went with unrelated edits - 16 x 4 registers through scratch memory, every
// access a vmcnt(0) (measured: the walk at L2 latency).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Per-wave fragment counts of the four contractions (a fragment = one MFMA "A" operand: 32 channels x 16 k)
template <int C, int CI>
struct Geo {
    static constexpr int KS_C = C / 16, KS_I = CI / 16;       // k-slices of a C- / CI-deep contraction
    static constexpr int MT_C = C / 128, MT_I = CI / 128;     // 32-channel tiles per wave of a C- / CI-wide layer
    static constexpr int TP = CI >= 256 ? 4 : 2;              // 32-channel tiles per ffn.0 pass
    static constexpr int NP = CI / (32 * TP);                 // ffn.0 passes (the wave's CI of the 4 CI channels)
    static constexpr int F_DC3 = MT_C * KS_I;                 // dc.3:  CI -> C
    static constexpr int F_FFN0 = NP * TP * KS_C;             // ffn.0: C -> 4 CI
    static constexpr int F_FFN2 = MT_C * KS_I;                // ffn.2: CI -> C
    static constexpr int F_MAIN = F_DC3 + F_FFN0 + F_FFN2;
    static constexpr int F_DC0 = MT_I * KS_C;                 // next dc.0: C -> CI
    static_assert(C % 128 == 0 && CI % 128 == 0, "channel counts in units of 4 waves x 32");
};

struct NsParams {
    const half_t* t2; int ldt;
    const half_t* x; int ldx;
    const half8* wmain;       // packed: [4 waves][F_MAIN][64 lanes]
    const half8* wnext;       // packed: [4 waves][F_DC0][64 lanes] or null
    const half_t* b3; const half_t* b0; const half_t* b2; const half_t* b1n;
    const half_t* q; const half_t* q2;
    const half_t* qf;         // 8-wave kernel, closing conv in the NEXT slot (wnext / b1n / t1n / ldt1 are then ITS weights, bias, output): its quant scale or null
    const float4* wsilu;
    half_t* y; int ldy;
    half_t* t1n; int ldt1;
    int M, shortcut;
    long long* timeline;      // optional [workgroups][32] shader-clock stamps of wave 0 (tools/probes/core_bench.hip)
};

// One LDS-DMA piece (64 lanes x 16 B, lane l lands at lds_dst + 16 l), wave-uniform base + 32-bit lane offset.
__device__ __forceinline__ void lds_dma16(const void* sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

template <int C, int CI, int PXT, bool NEXT>
__global__ void __launch_bounds__(NTHREADS, 1)
dcb_nsplit_kernel(const NsParams p)
{
    using G = Geo<C, CI>;
    // 512-wide layers x 64 pixels: 8 accumulator tiles per layer + 2 x 8 per ffn.0 pass leave room for 8 fragments in flight
    constexpr int RING = (C == 512 && CI == 512 && PXT == 2) ? 8 : RING_DEFAULT;
    constexpr int PX = 32 * PXT;
    constexpr int KS_C = G::KS_C, KS_I = G::KS_I, MT_C = G::MT_C, MT_I = G::MT_I, NP = G::NP, TP = G::TP;
    constexpr int CH_C = C / 8, CH_I = CI / 8;                  // 16-byte chunks per row
    constexpr int PITCH_C = C * 2, PITCH_I = CI * 2;
    // A: [PX][CI] (t2, then t); B: [PX][C] (y1, then y as dc.0's operand)
    constexpr int BUF_A = PX * PITCH_I, BUF_B = PX * PITCH_C;
    constexpr int OFF_B = BUF_A;
    constexpr int OFF_TABLE = align16k(BUF_A + BUF_B);
    static_assert(OFF_TABLE % 16384 == 0, "the WSiLU table must sit at a multiple of 16 KB (arith.h wsilu_row_lds)");
    constexpr int OFF_BIAS = OFF_TABLE + R * TABLE_BYTES;           // fp32: b3 (C) | b0 (4 CI) | b2 (C) | b1n (CI)
    constexpr int BIAS_FLOATS = 2 * C + 5 * CI;
    constexpr int OFF_Q = OFF_BIAS + BIAS_FLOATS * 4;               // fp16: q | q2
    constexpr int TOTAL = G::F_MAIN + (NEXT ? G::F_DC0 : 0);
#ifndef NS_DIRECT
#define NS_DIRECT 3
#endif
    // Output rows (bit 0: t1', bit 1: y) are stored straight from the epilogues' registers - 16 bytes per lane, half-waves
    // pairing up to 32-byte pieces of a row, L2 merges the pieces of a line - instead of staged in LDS, synchronised and
    // copied out as whole rows: a barrier and a 2 k-cycle copy less per output (A/B on one box, round 3:
    // intra 80.8 -> 83.8, HT-S 430 -> 443 pictures/s on a throttled box; every shape of the block bench equal or faster).
    // NS_DIRECT=0 builds the staged form.
    constexpr bool DIRECT_T1 = (NS_DIRECT & 1) != 0, DIRECT_Y = (NS_DIRECT & 2) != 0;
    static_assert((PX * CH_C) % NTHREADS == 0 && (PX * CH_I) % NTHREADS == 0, "tile rows must split evenly over the threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bufA = smem;
    char* const bufB = smem + OFF_B;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31;
    const int hi = lane >> 5;
    const int ntiles = (p.M + PX - 1) / PX;
    int tile = blockIdx.x;                               // persistent: tiles blockIdx.x, + gridDim.x, ...
    int m0 = tile * PX;
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) void*)smem));
    if ((lds_base & 16383u) != 0) __builtin_trap();      // dynamic LDS starts at 0 (no static LDS in this kernel)
    // stamps (first tile only): 0 entry | 1 constants + first t2 in LDS | 2 dc.3 MFMAs + x | 3 dc.3 epilogue | one per
    // ffn.0 pass (MFMAs + the previous pass's epilogue) | last epilogue | ffn.2 MFMAs | epilogue | y out | dc.0 MFMAs |
    // epilogue | t1' out
    const long long rt0 = static_cast<long long>(__builtin_amdgcn_s_memrealtime());
    int stamp_no = 0;
    auto stamp = [&]() {
        if (p.timeline != nullptr && tid == 0 && stamp_no < 29 && tile == static_cast<int>(blockIdx.x)) {
            p.timeline[static_cast<size_t>(blockIdx.x) * 32 + stamp_no] = static_cast<long long>(__builtin_readcyclecounter());
        }
        ++stamp_no;
    };
    stamp();

    // ---- L2 warm-up (dcb_core.hip: every workgroup streams the SAME weights at the same time and L2 starts cold
    // at a kernel boundary; each workgroup first touches ITS share of the stream, all misses in flight together)
    unsigned warm = 0;
    {
        const int rank = (blockIdx.x >> 3) & 31;
        constexpr int LINES = 4 * G::F_MAIN * 8;                      // 128-byte lines of the main stream
#pragma unroll
        for (int k = 0; k < (LINES + 32 * NTHREADS - 1) / (32 * NTHREADS); ++k) {
            const int ql = rank + 32 * (tid + NTHREADS * k);
            if (ql < LINES) warm ^= *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(p.wmain) + static_cast<size_t>(ql) * 128);
        }
    }

    // ---- constants: loaded into registers FIRST (16-byte units, all loads independent). Every use of a loaded
    // register waits for everything issued before it (vmcnt retires in order): round 3's first version read the
    // biases one value per loop iteration behind the tile transfers - 13 dependent memory round trips, 19.6 k cycles
    // of prologue per workgroup (profiles/r03_nsplit_ablation.txt).
    constexpr int TAB_PER_THREAD = R * WSILU_SEGMENTS / NTHREADS;
    static_assert(TAB_PER_THREAD == 4, "four table rows per thread, spelled out (as a loop the array went through scratch)");
    const float4 tab0 = p.wsilu[tid / R], tab1 = p.wsilu[(tid + NTHREADS) / R], tab2 = p.wsilu[(tid + 2 * NTHREADS) / R],
                 tab3 = p.wsilu[(tid + 3 * NTHREADS) / R];
    constexpr int CONST_UNITS = (BIAS_FLOATS + 2 * C) / 8;      // b3 | b0 | b2 | b1n | q | q2 in 8-channel units
    constexpr int CONST_PER_THREAD = (CONST_UNITS + NTHREADS - 1) / NTHREADS;
    static_assert(CONST_PER_THREAD <= 4, "four named registers below");
    auto const_unit = [&](int k) {
        const int ch = min(tid + k * NTHREADS, CONST_UNITS - 1) * 8;
        const half_t* src = ch < C ? p.b3 + ch
                          : ch < C + 4 * CI ? p.b0 + (ch - C)
                          : ch < 2 * C + 4 * CI ? p.b2 + (ch - C - 4 * CI)
                          : ch < BIAS_FLOATS ? (p.b1n != nullptr ? p.b1n + (ch - 2 * C - 4 * CI) : p.b2)
                          : ch < BIAS_FLOATS + C ? (p.q != nullptr ? p.q + (ch - BIAS_FLOATS) : p.b2)
                          : (p.q2 != nullptr ? p.q2 + (ch - BIAS_FLOATS - C) : p.b2);
        return *reinterpret_cast<const half8*>(src);
    };
    const half8 cv0 = const_unit(0), cv1 = const_unit(CONST_PER_THREAD > 1 ? 1 : 0), cv2 = const_unit(CONST_PER_THREAD > 2 ? 2 : 0),
                cv3 = const_unit(CONST_PER_THREAD > 3 ? 3 : 0);
    const float* const lb3 = reinterpret_cast<const float*>(smem + OFF_BIAS);
    const float* const lb0 = lb3 + C;
    const float* const lb2 = lb3 + C + 4 * CI;
    const float* const lb1n = lb3 + 2 * C + 4 * CI;
    const half_t* const lq = reinterpret_cast<const half_t*>(smem + OFF_Q);
    const half_t* const lq2 = lq + C;
    unsigned tab = lds_base + OFF_TABLE + (lane & (R - 1)) * 16;

    // ---- the wave's weight stream: fragment f at byte offset f * 1024 (one contiguous KB per fragment, lane-linear),
    // as 32-bit byte offsets from the (kernel-argument, hence provably global) stream pointers: the loads then take
    // the scalar-base + lane-offset form of global_load. (A laundered POINTER loses its address space: flat_load, which
    // counts on lgkmcnt as well and turns every counted wait into vmcnt(0) - measured: the walk 50 % slower.)
    unsigned wsm = static_cast<unsigned>(wave * G::F_MAIN * 64 + lane) * 16u;
    unsigned wsn = static_cast<unsigned>(wave * G::F_DC0 * 64 + lane) * 16u;
    // (NS_BUFLOAD: the streams as buffer resources - base, no stride, no bounds in the way, raw dword format)
    const __amdgpu_buffer_rsrc_t rs_main = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8*>(p.wmain), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8*>(NEXT ? p.wnext : p.wmain), 0, 0x7fffffff, 0x00020000);
    // Fragment f of the stream lives in ring[f % RING] from its load (issued RING fragments ahead) to its MFMAs
    half8 ring[RING];
    // Ablation switches (tools/build_variant.sh; RESULTS ARE WRONG with any of them): NS_EXP_NOLOAD = no weight loads
    // behind the first RING fragments, NS_EXP_NOEPI = ffn.0's WSiLU + chunk sum replaced by a plain conversion
    auto issue = [&](auto f_tag) {
        constexpr int f = decltype(f_tag)::value;
#ifdef NS_EXP_NOLOAD
        if constexpr (f >= RING) return;
#endif
        if constexpr (f < G::F_MAIN) {
            if constexpr (BUFLOAD) {
                ring[f % RING] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_main, wsm + static_cast<unsigned>(f & 3) * 1024u, (f >> 2) * 4096, 0));
            } else {
                ring[f % RING] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(p.wmain) + (wsm + static_cast<unsigned>(f) * 1024u));
            }
        } else if constexpr (f < TOTAL) {
            constexpr int g = f - G::F_MAIN;
            if constexpr (BUFLOAD) {
                ring[f % RING] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_next, wsn + static_cast<unsigned>(g & 3) * 1024u, (g >> 2) * 4096, 0));
            } else {
                ring[f % RING] = *reinterpret_cast<const half8*>(reinterpret_cast<const char*>(p.wnext) + (wsn + static_cast<unsigned>(g) * 1024u));
            }
        }
    };
    // ---- a [PX][W] tile of whole rows, memory -> LDS by LDS-DMA; LDS image lane-linear, the bank swizzle (16-byte
    // chunk c of row r lives at chunk c ^ (r & 15)) sits on the SOURCE side. Rows behind the picture read its last row.
    int tidv = tid;                  // (made opaque once per tile: see the loop head)
    auto dma_tile = [&](auto chunks_tag, const half_t* base, int ld, unsigned lds_off, int first_row) {
        constexpr int CHN = decltype(chunks_tag)::value;
        const int f = min(first_row, p.M - 1);
        const int last = p.M - 1 - f;
        const half_t* const w = base + static_cast<size_t>(f) * ld;
#pragma unroll
        for (int i = 0; i < PX * CHN / NTHREADS; ++i) {
            const int pos = i * NTHREADS + tidv;
            const int r = pos / CHN, pc = pos % CHN;
            const int lc = pc ^ (r & 15);
            const int rr = min(r, last);
            lds_dma16(w, static_cast<unsigned>(rr * ld + lc * 8) * 2u, lds_base + lds_off + (i * NTHREADS + wave * 64) * 16);
        }
    };
    using ChI = std::integral_constant<int, CH_I>;
    using ChC = std::integral_constant<int, CH_C>;
    // ---- the block input x of a tile (dc.3's residual) waits in registers, in the layout of dc.3's epilogue: 16-byte
    // runs of this lane's pixel
    half8 xr[MT_C][PXT][2];
    int pxv = px, hiv = hi;          // (made opaque once per tile: see the loop head)
    auto load_x = [&](int first_row) {
#pragma unroll
        for (int t = 0; t < PXT; ++t) {
            const half_t* const row = p.x + static_cast<size_t>(min(first_row + 32 * t + pxv, p.M - 1)) * p.ldx + (32 * MT_C * wave + 8 * hiv);
#pragma unroll
            for (int j = 0; j < MT_C; ++j)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) xr[j][t][pr] = *reinterpret_cast<const half8*>(row + 32 * j + 16 * pr);
        }
    };
    // first tile: t2 -> A, then the first weight fragments, then x (dc.3's epilogue is its first use): everything in
    // front of x (the constants included) is waited for here
    dma_tile(ChI{}, p.t2, p.ldt, 0, m0);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, RING>([&](auto i) { issue(i); });
    __builtin_amdgcn_sched_barrier(0);
    load_x(m0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT_C * PXT * 2) : "memory");
    {
        float4* t = reinterpret_cast<float4*>(smem + OFF_TABLE);
        t[tid] = tab0; t[tid + NTHREADS] = tab1; t[tid + 2 * NTHREADS] = tab2; t[tid + 3 * NTHREADS] = tab3;
        float* lb = reinterpret_cast<float*>(smem + OFF_BIAS);
        half_t* lqw = reinterpret_cast<half_t*>(smem + OFF_Q);
        auto put = [&](int k, const half8 v) {
            const int u = tid + k * NTHREADS;
            if (u < BIAS_FLOATS / 8) {
                float4v lo4, hi4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo4[e] = static_cast<float>(v[e]);
                    hi4[e] = static_cast<float>(v[4 + e]);
                }
                *reinterpret_cast<float4v*>(lb + u * 8) = lo4;
                *reinterpret_cast<float4v*>(lb + u * 8 + 4) = hi4;
            } else if (u < CONST_UNITS) {
                *reinterpret_cast<half8*>(lqw + (u - BIAS_FLOATS / 8) * 8) = v;
            }
        };
        put(0, cv0);
        if (CONST_PER_THREAD > 1) put(1, cv1);
        if (CONST_PER_THREAD > 2) put(2, cv2);
        if (CONST_PER_THREAD > 3) put(3, cv3);
    }
    __syncthreads();
    if (warm == 0x9e3779b9u && p.M < 0) p.y[0] = static_cast<half_t>(0);     // never true: keeps the warm-up loads alive
    stamp();

    // ---- fragment addressing. Row of pixel tile t: (32 t + px) * pitch; chunk c of a row sits at c ^ (px & 15).
    // B fragment of k-slice ks: chunk 2 ks + hi = (2 ks) ^ hi, so the lane part of the swizzle is one constant.
    int s0 = (hi ^ (px & 15)) << 4;
    int rowA = px * PITCH_I, rowB = px * PITCH_C + OFF_B;     // byte offsets from smem
    int hi4 = 4 * hi;                 // (bias_tile)
    // (NS_ADDR8: (32 ks) ^ s0 = ((32 (ks & 7)) ^ s0) + 256 (ks >> 3): s0 has bits 4 .. 7 only)
    int fa8[8], fb8[8];
    auto frag_a = [&](int t, int ks) {
        if constexpr (ADDR8) return *reinterpret_cast<const half8*>(smem + fa8[ks & 7] + (t * (32 * PITCH_I) + (ks >> 3) * 256));
        else return *reinterpret_cast<const half8*>(smem + rowA + t * (32 * PITCH_I) + ((ks * 32) ^ s0));
    };
    auto frag_b = [&](int t, int ks) {
        if constexpr (ADDR8) return *reinterpret_cast<const half8*>(smem + fb8[ks & 7] + (t * (32 * PITCH_C) + (ks >> 3) * 256));
        else return *reinterpret_cast<const half8*>(smem + rowB + t * (32 * PITCH_C) + ((ks * 32) ^ s0));
    };
    // the 16-byte run of channels ch0 + 8 hi .. + 7 (ch0 a multiple of 16) of this lane's pixel in tile t
    auto run_a = [&](int t, int ch0) { return reinterpret_cast<half8*>(smem + rowA + t * (32 * PITCH_I) + ((ch0 * 2) ^ s0)); };
    auto run_b = [&](int t, int ch0) { return reinterpret_cast<half8*>(smem + rowB + t * (32 * PITCH_C) + ((ch0 * 2) ^ s0)); };
    // accumulator tile (32 channels from `first`) initialised with the bias: acc[r] = channel first + 8 (r>>2) + 4 hi + (r&3)
    auto bias_tile = [&](float16v& acc, const float* bias, int first) {
        const float* bp = bias + first + hi4;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4v b4 = *reinterpret_cast<const float4v*>(bp + 8 * g4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g4 + e] = b4[e];
        }
    };
    // accumulator tile -> run pr: channels 16 pr + 8 hi .. + 7 of the tile, this lane's pixel (half-waves paired up)
    auto runs_of = [&](const float16v& a, int pr, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * pr + e]), __float_as_uint(a[8 * pr + 4 + e]), false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
        }
    };
    // NT tiles x PXT pixel tiles over KSN k-slices, activations through `frag` (A or B); F0 = stream index of the
    // contraction's first fragment; `piece(ks)`: work of an EARLIER accumulator set (an epilogue, cut into pieces)
    // issued beside the MFMAs of slice ks
    auto contract = [&](auto nt_tag, auto ks_tag, auto f0_tag, auto&& frag, auto& acc, auto&& piece) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr int KSN = decltype(ks_tag)::value;
        constexpr int F0 = decltype(f0_tag)::value;
        half8 b[2][PXT];              // activation fragments, read one k-slice ahead of their MFMAs
#pragma unroll
        for (int t = 0; t < PXT; ++t) b[0][t] = frag(t, 0);
        static_for<0, KSN>([&](auto kt) {
            constexpr int ks = decltype(kt)::value;
            if constexpr (ks + 1 < KSN) {
#pragma unroll
                for (int t = 0; t < PXT; ++t) b[(ks + 1) & 1][t] = frag(t, ks + 1);
            }
            __builtin_amdgcn_sched_barrier(0);       // ... and stay in front of this slice's MFMAs
            static_for<0, NT>([&](auto j_tag) {
                constexpr int j = WAITLAST ? NT - 1 - decltype(j_tag)::value : decltype(j_tag)::value;
                const half8 a = ring[(F0 + ks * NT + j) % RING];
#pragma unroll
                for (int t = 0; t < PXT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[ks & 1][t], acc[j][t], 0, 0, 0);
            });
            piece(ks);
            static_for<0, NT>([&](auto j_tag) { issue(std::integral_constant<int, F0 + ks * NT + decltype(j_tag)::value + RING>{}); });
            // nothing crosses a k-slice: left alone, hipcc sinks every prefetch load down to the MFMA that consumes it
            // (register pressure) and waits vmcnt(0) right behind it - the whole stream then runs at L2 latency
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using TagMTC = std::integral_constant<int, MT_C>;
    using TagMTI = std::integral_constant<int, MT_I>;
    using TagTP = std::integral_constant<int, TP>;
    using KsC = std::integral_constant<int, KS_C>;
    using KsI = std::integral_constant<int, KS_I>;
    auto no_piece = [](int) {};

    // whole rows of an LDS tile -> memory, 16 bytes per lane, consecutive lanes = consecutive chunks of a row
    auto copy_out = [&](auto chunks_tag, const char* buf, int pitch, half_t* dst, int ld) {
        constexpr int CHN = decltype(chunks_tag)::value;
#pragma unroll
        for (int i = 0; i < PX * CHN / NTHREADS; ++i) {
            const int pos = i * NTHREADS + tidv;
            const int r = pos / CHN, lc = pos % CHN;
            const half8 v = *reinterpret_cast<const half8*>(buf + r * pitch + ((lc ^ (r & 15)) << 4));
            if (m0 + r < p.M) store_line(dst + static_cast<size_t>(m0 + r) * ld + lc * 8, v);
        }
    };

    // ================================================================ persistent loop over this workgroup's tiles
    // (tile, tile + gridDim.x, ...). The constants above are loaded once; t2 of the NEXT tile is requested right behind
    // ffn.2's MFMAs of this one (into A, dead by then), its x (into registers) behind dc.0's MFMAs - each in front of an
    // epilogue + row copy of 7 - 8 k cycles: only the first tile's transfers are exposed (as a kernel of one tile per workgroup
    // the prologue was 12 k of 87 k cycles, profiles/r03_nsplit_ablation.txt). The PLACE matters: memory operations
    // retire in order, so a transfer from HBM / the Infinity Cache issued in front of a contraction holds up every weight
    // fragment (an L2 hit) requested behind it - with t2 in front of dc.0 and x in front of dc.3 those two contractions
    // took 11.4 k and 7.9 k cycles instead of 5.7 k (profiles/r03_core_bench_dual0.txt).
    for (;;) {
    // Everything the unrolled body addresses hangs off these few per-lane values. Made opaque once per tile: as loop
    // invariants the compiler hoists EVERY derived address out of the loop (one register pair per weight fragment,
    // one register per LDS fragment: 1 000+ values) and spills them all (measured: 1 023 spilled registers).
    asm volatile("" : "+v"(wsm), "+v"(wsn), "+v"(tab), "+v"(s0), "+v"(rowA), "+v"(rowB), "+v"(hi4), "+v"(tidv), "+v"(pxv), "+v"(hiv));
    if constexpr (ADDR8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa8[i] = rowA + ((32 * i) ^ s0);
            fb8[i] = rowB + ((32 * i) ^ s0);
        }
    }
    const int next_tile = tile + static_cast<int>(gridDim.x);
    const bool has_next = next_tile < ntiles;
    // ================================================================ dc.3: y1 = W3 t2 + b3' + x   (A -> B)
    {
        float16v acc[MT_C][PXT];
        auto dc3_run = [&](auto j_tag, auto t_tag, auto pr_tag) {
            constexpr int j = decltype(j_tag)::value, t = decltype(t_tag)::value, pr = decltype(pr_tag)::value;
            float v[8];
            runs_of(acc[j][t], pr, v);
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = to_half(v[e] + static_cast<float>(xr[j][t][pr][e]));
            *run_b(t, 32 * (wave * MT_C + j) + 16 * pr) = o;
        };
#pragma unroll
        for (int j = 0; j < MT_C; ++j)
#pragma unroll
            for (int t = 0; t < PXT; ++t) bias_tile(acc[j][t], lb3, 32 * (wave * MT_C + j));
        contract(TagMTC{}, KsI{}, std::integral_constant<int, 0>{}, frag_a, acc, no_piece);
        stamp();
        static_for<0, MT_C>([&](auto j_tag) {
            static_for<0, PXT>([&](auto t_tag) {
                dc3_run(j_tag, t_tag, std::integral_constant<int, 0>{});
                dc3_run(j_tag, t_tag, std::integral_constant<int, 1>{});
            });
        });
    }
    __syncthreads();            // y1 complete in B; every wave is done with t2 in A
    stamp();

    // ================================================================ ffn.0: t = chunk_add(WSiLU(W0 y1 + b0))   (B -> A)
    // The wave's CI ffn.0 channels in NP passes of TP tiles: a pass = 32 TP ffn.0 channels = 8 TP channels of t.
    // The WSiLU / chunk-sum epilogue of a pass (32 TP values per lane: 6 VALU operations and one 16-byte table gather
    // each) runs UNDER THE MFMAs OF THE NEXT PASS, cut into 2 TP PXT half-tiles of 8 values, each in two stages one
    // k-slice apart (rows gathered | polynomials + sums): serial, it was 18 % of the kernel (5.5 k of 13.6 k cycles
    // per pass, profiles/r03_nsplit_ablation.txt). Two accumulator sets alternate; only the last pass's epilogue
    // is exposed.
    {
        constexpr int NHC = 2 * TP * PXT;                 // half-tiles of a pass: (t, np, h, half)
        constexpr bool PIPE = NP > 1;
        float16v accs[PIPE ? 2 : 1][TP][PXT];
        float4v crow[8];                                  // (plain vectors: an array of float4 structs went through scratch)
        float sums[2][4];                                 // [h][g] of the (t, np) pair being finished
        // stage A of half-tile i runs beside the MFMAs of slice_of_hc(i) - one half-tile per slice: the row registers -,
        // stage B one slice later; with as many half-tiles as slices the last stage B follows the contraction
        static_assert(NHC <= KS_C, "one k-slice per half-tile of the previous pass");
        constexpr int SPAN = NHC <= KS_C - 1 ? KS_C - 1 : KS_C;
        auto slice_of_hc = [&](int i) { return i * SPAN / NHC; };
        // stage A: table rows of the 8 values
        auto stage_a = [&](const float16v (&a)[TP][PXT], int i) {
            const int t = i / (2 * TP), np = (i / 4) % (TP / 2), h = (i / 2) % 2, half = i % 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 r = wsilu_row_lds<R, true>(a[2 * np + h][t][8 * half + e], tab);
                crow[e] = float4v{r.x, r.y, r.z, r.w};
            }
        };
        // lower half-wave collects the 8 outputs of tile 2 np, upper half-wave those of tile 2 np + 1
        auto write_pair = [&](int t, int np, int f0) {
            half8 o;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sums[0][g]), __float_as_uint(sums[1][g]), false, false);
                o[2 * g] = to_half(__uint_as_float(sw[0]));
                o[2 * g + 1] = to_half(__uint_as_float(sw[1]));
            }
            *run_a(t, (f0 + 64 * np) / 4) = o;                 // t channels (f0 + 64 np) / 4 + 8 hi .. + 7
        };
        // stage B: polynomials, chunk sums; behind the last half-tile of a (t, np) pair the 8 outputs go to A
        auto stage_b = [&](const float16v (&a)[TP][PXT], int i, int f0) {
            const int t = i / (2 * TP), np = (i / 4) % (TP / 2), h = (i / 2) % 2, half = i % 2;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                auto row = [&](int e) { const float4v r = crow[4 * g + e]; return make_float4(r[0], r[1], r[2], r[3]); };
                float s = a[2 * np + h][t][8 * half + 4 * g] * wsilu_poly(a[2 * np + h][t][8 * half + 4 * g], row(0));
#pragma unroll
                for (int e = 1; e < 4; ++e) {
                    s = fmaf(a[2 * np + h][t][8 * half + 4 * g + e], wsilu_poly(a[2 * np + h][t][8 * half + 4 * g + e], row(e)), s);
                }
                sums[h][2 * half + g] = s;
            }
            if (h == 1 && half == 1) write_pair(t, np, f0);
        };
        // one piece of the previous pass's epilogue beside the MFMAs of slice ks: arithmetic of half-tile i, THEN the
        // gathers of half-tile i + 1 into the same 8 row registers (two live row sets - gathers first - pushed the
        // kernel to all 512 registers and the walk from 11.7 k to 18.3 k cycles per pass)
        auto epilogue_piece = [&](const float16v (&a)[TP][PXT], int f0, int ks) {
#ifndef NS_EXP_NOEPI
#pragma unroll
            for (int i = 0; i < NHC; ++i) {
                if (slice_of_hc(i) + 1 == ks) stage_b(a, i, f0);
            }
#pragma unroll
            for (int i = 0; i < NHC; ++i) {
                if (slice_of_hc(i) == ks) stage_a(a, i);
            }
#endif
        };
        // the whole epilogue of a pass with nothing to hide behind: one accumulator tile at a time, all 16 gathers of
        // a tile in flight before its polynomials
        auto epilogue_serial = [&](const float16v (&a)[TP][PXT], int f0) {
#pragma unroll
            for (int t = 0; t < PXT; ++t)
#pragma unroll
                for (int np = 0; np < TP / 2; ++np) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#ifdef NS_EXP_NOEPI
#pragma unroll
                        for (int g = 0; g < 4; ++g) sums[h][g] = a[2 * np + h][t][4 * g];
#else
                        constexpr int GB = 16;        // gathers in flight
#pragma unroll
                        for (int g0 = 0; g0 < 16; g0 += GB) {
                            float4v c[GB];
#pragma unroll
                            for (int e = 0; e < GB; ++e) {
                                const float4 r = wsilu_row_lds<R, true>(a[2 * np + h][t][g0 + e], tab);
                                c[e] = float4v{r.x, r.y, r.z, r.w};
                            }
#pragma unroll
                            for (int g = 0; g < GB / 4; ++g) {
                                auto row = [&](int e) { const float4v r = c[4 * g + e]; return make_float4(r[0], r[1], r[2], r[3]); };
                                const int v0 = g0 + 4 * g;
                                float s = a[2 * np + h][t][v0] * wsilu_poly(a[2 * np + h][t][v0], row(0));
#pragma unroll
                                for (int e = 1; e < 4; ++e) s = fmaf(a[2 * np + h][t][v0 + e], wsilu_poly(a[2 * np + h][t][v0 + e], row(e)), s);
                                sums[h][g0 / 4 + g] = s;
                            }
                        }
#endif
                    }
                    write_pair(t, np, f0);
                }
        };
        static_for<0, NP>([&](auto pass_tag) {
            constexpr int pass = decltype(pass_tag)::value;
            using F0 = std::integral_constant<int, G::F_DC3 + pass * TP * KS_C>;
            const int f0 = wave * CI + pass * (32 * TP);        // first ffn.0 channel of the pass
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int t = 0; t < PXT; ++t) bias_tile(accs[PIPE ? pass & 1 : 0][j][t], lb0, f0 + 32 * j);
            if constexpr (!PIPE) {
                contract(TagTP{}, KsC{}, F0{}, frag_b, accs[0], no_piece);
                stamp();
                if constexpr (pass + 1 < NP) epilogue_serial(accs[0], f0);
            } else if constexpr (pass == 0) {
                contract(TagTP{}, KsC{}, F0{}, frag_b, accs[pass & 1], no_piece);
                stamp();
            } else {
                contract(TagTP{}, KsC{}, F0{}, frag_b, accs[pass & 1],
                         [&](int ks) { epilogue_piece(accs[(pass - 1) & 1], f0 - 32 * TP, ks); });
                if constexpr (SPAN == KS_C) epilogue_piece(accs[(pass - 1) & 1], f0 - 32 * TP, KS_C);
                stamp();
            }
        });
        // the last pass's epilogue has nothing to hide behind (and needs no weights: the next tile's x goes out in front of it)
        if constexpr (XEARLY == 1) {
            load_x(next_tile * PX);
            __builtin_amdgcn_sched_barrier(0);
        }
        epilogue_serial(accs[PIPE ? (NP - 1) & 1 : 0], wave * CI + (NP - 1) * (32 * TP));
        stamp();
    }
    __syncthreads();            // t complete in A; every wave is done with y1 as an operand

    // ================================================================ ffn.2: y = (W2 t + b2 + y1 [+ x]) [* q] -> fp16 [* q2]   (A -> B in place of y1)
    {
        float16v acc[MT_C][PXT];
        auto ffn2_run = [&](auto j_tag, auto t_tag, auto pr_tag) {
            constexpr int j = decltype(j_tag)::value, t = decltype(t_tag)::value, pr = decltype(pr_tag)::value;
            const int ch = 32 * (wave * MT_C + j) + 16 * pr;          // + 8 hi
            float v[8];
            runs_of(acc[j][t], pr, v);
            half8* const slot = run_b(t, ch);
            const half8 y1 = *slot;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(y1[e]);
            if (p.shortcut) {
                const int m = min(m0 + 32 * t + px, p.M - 1);
                const half8 r8 = *reinterpret_cast<const half8*>(p.x + static_cast<size_t>(m) * p.ldx + ch + 8 * hi);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(r8[e]);
            }
            if (p.q != nullptr) {
                const half8 q8 = *reinterpret_cast<const half8*>(lq + ch + 8 * hi);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * static_cast<float>(q8[e]);
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
            if (p.q2 != nullptr) {
                const half8 q8 = *reinterpret_cast<const half8*>(lq2 + ch + 8 * hi);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = hmul(o[e], q8[e]);
            }
            if constexpr (NEXT || !DIRECT_Y) *slot = o;          // dc.0's operand
            if constexpr (DIRECT_Y) {
                // 16 bytes per lane straight from the registers (half-waves pair up to 32-byte pieces of a row)
                const int m = m0 + 32 * t + pxv;
                if (m < p.M) store_line(p.y + static_cast<size_t>(m) * p.ldy + ch + 8 * hiv, o);
            }
        };
        // the next tile's transfers (see the loop head): t is dead once every wave is behind its last ffn.2 MFMA
        auto behind_mfmas = [&] {
            stamp();
            __syncthreads();
            if (has_next) dma_tile(ChI{}, p.t2, p.ldt, 0, next_tile * PX);
            // (unconditional - rows are clamped to the picture -: a conditional load keeps the old values alive)
            if constexpr ((!NEXT && XEARLY == 0) || XEARLY == 2) load_x(next_tile * PX);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int j = 0; j < MT_C; ++j)
#pragma unroll
            for (int t = 0; t < PXT; ++t) bias_tile(acc[j][t], lb2, 32 * (wave * MT_C + j));
        contract(TagMTC{}, KsI{}, std::integral_constant<int, G::F_DC3 + G::F_FFN0>{}, frag_a, acc, no_piece);
        behind_mfmas();
        static_for<0, MT_C>([&](auto j_tag) {
            static_for<0, PXT>([&](auto t_tag) {
                ffn2_run(j_tag, t_tag, std::integral_constant<int, 0>{});
                ffn2_run(j_tag, t_tag, std::integral_constant<int, 1>{});
            });
        });
    }
    if constexpr (NEXT || !DIRECT_Y) __syncthreads();            // y complete in B; every wave is done with t in A
    stamp();
    if constexpr (!DIRECT_Y) copy_out(ChC{}, bufB, PITCH_C, p.y, p.ldy);
    stamp();

    // ================================================================ dc.0 of the next block: t1' = WSiLU(W1' y + b1')   (B -> B)
    if constexpr (NEXT) {
        float16v acc[MT_I][PXT];
        auto dc0_run = [&](auto j_tag, auto t_tag, auto pr_tag) {
            constexpr int j = decltype(j_tag)::value, t = decltype(t_tag)::value, pr = decltype(pr_tag)::value;
            float v[8];
            runs_of(acc[j][t], pr, v);
            float4v c[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 r = wsilu_row_lds<R, true>(v[e], tab);
                c[e] = float4v{r.x, r.y, r.z, r.w};
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = to_half(v[e] * wsilu_poly(v[e], make_float4(c[e][0], c[e][1], c[e][2], c[e][3])));
            if constexpr (DIRECT_T1) {
                const int m = m0 + 32 * t + pxv;
                if (m < p.M) store_line(p.t1n + static_cast<size_t>(m) * p.ldt1 + 32 * (wave * MT_I + j) + 16 * pr + 8 * hiv, o);
            } else {
                // staged with B's pitch, the CI channels in the first CI / 8 chunks of a row
                *run_b(t, 32 * (wave * MT_I + j) + 16 * pr) = o;
            }
        };
        auto behind_mfmas = [&] {
            // the ring is empty: the first fragments of the next tile go out now and arrive under the epilogue below
            if (has_next) {
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, RING>([&](auto i) { issue(i); });
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp();
            if constexpr (!DIRECT_T1) __syncthreads();        // every wave is done with y as an operand (and with copying it out): B becomes the staging area
            // x of the next tile: a second burst of its own (all CUs ask at the same moment: 12 MB at 1080p), under this epilogue
            if constexpr (XEARLY == 0) load_x(next_tile * PX);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int j = 0; j < MT_I; ++j)
#pragma unroll
            for (int t = 0; t < PXT; ++t) bias_tile(acc[j][t], lb1n, 32 * (wave * MT_I + j));
        contract(TagMTI{}, KsC{}, std::integral_constant<int, G::F_MAIN>{}, frag_b, acc, no_piece);
        behind_mfmas();
        static_for<0, MT_I>([&](auto j_tag) {
            static_for<0, PXT>([&](auto t_tag) {
                dc0_run(j_tag, t_tag, std::integral_constant<int, 0>{});
                dc0_run(j_tag, t_tag, std::integral_constant<int, 1>{});
            });
        });
        if constexpr (!DIRECT_T1) __syncthreads();
        stamp();
        if constexpr (!DIRECT_T1) copy_out(ChI{}, bufB, PITCH_C, p.t1n, p.ldt1);
        stamp();
    } else {
        if (has_next) {
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, RING>([&](auto i) { issue(i); });
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!has_next) break;
    // the next tile's t2 has landed (every wave waits for its own pieces, the barrier covers the others'), the rows
    // copied out of B are read: B is free for the next tile's y1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tile = next_tile;
    m0 = tile * PX;
    }       // tiles
    // stamp 31: the workgroup's last instruction (all tiles): shader cycles of the whole launch per workgroup; stamps 29 / 30: the
    // constant 100 MHz clock (s_memrealtime) at entry / here: cycles / time = the shader clock the launch really ran at
    if (p.timeline != nullptr && tid == 0) {
        p.timeline[static_cast<size_t>(blockIdx.x) * 32 + 31] = static_cast<long long>(__builtin_readcyclecounter());
        p.timeline[static_cast<size_t>(blockIdx.x) * 32 + 30] = static_cast<long long>(__builtin_amdgcn_s_memrealtime());
        p.timeline[static_cast<size_t>(blockIdx.x) * 32 + 29] = rt0;
    }
}
template <int C, int CI, int PXT>
constexpr int smem_bytes()
{
    return align16k(32 * PXT * (C + CI) * 2) + R * TABLE_BYTES + (2 * C + 5 * CI) * 4 + 2 * C * 2;
}

template <int C, int CI, int PXT, bool NEXT>
void launch(const NsParams& p, hipStream_t stream)
{
    auto kern = dcb_nsplit_kernel<C, CI, PXT, NEXT>;
    constexpr int smem = smem_bytes<C, CI, PXT>();
    static_assert(smem <= 160 * 1024, "LDS budget");
    // per device (advisor, round 3: a process-wide once-flag left a second device without the LDS attribute and with the
    // first device's CU count): the attribute is set, and the CU count read, once per device id
    constexpr int MAX_DEVICES = 64;
    static std::once_flag once[MAX_DEVICES];
    static int cu_count[MAX_DEVICES];
    int dev = 0;
    hip_check(hipGetDevice(&dev), "hipGetDevice");
    if (dev < 0 || dev >= MAX_DEVICES) throw std::runtime_error("dcb_nsplit: device id out of range");
    std::call_once(once[dev], [&] {
        hipDeviceProp_t prop;
        hip_check(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
            throw std::runtime_error(std::string("dcb_nsplit needs gfx950 (160 KB LDS, permlane32_swap); device is ") + prop.gcnArchName);
        }
        hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem),
                  "hipFuncSetAttribute(dcb_nsplit)");
        int n = 0;
        hip_check(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute");
        cu_count[dev] = n > 0 ? n : 256;
    });
    // persistent workgroups: one per CU (up to 160 KB of LDS each), tiles dealt round-robin
    const int cus = cu_count[dev];
    const int tiles = (p.M + 32 * PXT - 1) / (32 * PXT);
    const int grid = tiles < cus ? tiles : cus;
    hipEvent_t ev0, ev1;
    // 2 * pixels * C * kflop = FLOPs of the launch: dc.3 CI + ffn.0 4 CI + ffn.2 CI (+ dc.0 CI) per output channel of width C
    const int kflop = (NEXT ? 7 : 6) * CI;
    if (gemm_profile_slot(GemmLaunchInfo{p.M, C, kflop, 0x40000000, 0.f}, &ev0, &ev1)) {
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), smem, stream, ev0, ev1, 0, p);
    } else {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), smem, stream, p);
    }
    hip_check(hipGetLastError(), "dcb_nsplit launch");
}

// every instantiation of one block shape: 64 / 32 pixels per workgroup (768-wide blocks have LDS for 32 only), with /
// without the next block's dc.0. (Two 32-pixel workgroups per CU instead of one of 64 - 256 registers per wave, no
// second accumulator set - were measured and are slower for every shape but (256, 128): 101.7 vs 94.8 us at (384, 384),
// profiles/r03_core_bench_dual{0,1}.txt; git history has the kernel switch.)
template <int C, int CI>
void run_shape(const NsParams& p, bool wide, bool next, hipStream_t stream)
{
    if constexpr (C < 768) {
        if (wide) { if (next) launch<C, CI, 2, true>(p, stream); else launch<C, CI, 2, false>(p, stream); return; }
    }
    if (next) launch<C, CI, 1, true>(p, stream); else launch<C, CI, 1, false>(p, stream);
}

// dcb_nsplit_<shape>.hip
void run_256_128(const NsParams& p, bool wide, bool next, hipStream_t stream);
void run_256_256(const NsParams& p, bool wide, bool next, hipStream_t stream);
void run_384_384(const NsParams& p, bool wide, bool next, hipStream_t stream);
void run_512_256(const NsParams& p, bool wide, bool next, hipStream_t stream);
void run_512_512(const NsParams& p, bool wide, bool next, hipStream_t stream);
void run_768_768(const NsParams& p, bool wide, bool next, hipStream_t stream);

}  // namespace nsplit
}  // namespace dcvc
