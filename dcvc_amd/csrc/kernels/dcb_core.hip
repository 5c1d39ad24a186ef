// dcb_core.hip - everything of a full-width DepthConvBlock (C = 384: the intra encoder / decoder,
// 74 % of DMCI's MACs) that follows its depthwise conv, in ONE launch:
//
//     y1 = W3 * t2 + b3' + x                          dc.3 (+ folded depthwise bias) + block input
//     t  = chunk_add(WSiLU(W0 * y1 + b0))             ffn.0   (4x expansion, never materialised)
//     y  = (W2 * t + b2 + y1 [+ x]) [* q] -> fp16 [* q2]     ffn.2 (+ block shortcut, + quant scales)
//     [t1' = WSiLU(W1' * y + b1')]                    dc.0 of the NEXT block of a chain (optional)
//
// Reference: DepthConvBlockProxy::forward, layers_proxy.cpp:71-101, runs these as 3 (4) CUTLASS
// launches with y1, t (and y as a GEMM operand) round-tripping through memory; conv_gemm.hip did
// the same up to round 1 (5 launches per block, ~300 MB of traffic for 50 MB of compulsory bytes:
// below the ~310 FLOP/B ridge of the MI355X). Here every intermediate stays in
// REGISTERS:
//
//   * a workgroup owns 128 pixels, a wave 32 of them and ALL channels (4 waves, one per SIMD, up
//     to 512 registers each). D^T = W * X^T as in conv_gemm.hip (weights = MFMA "A" operand from
//     LDS, activations = "B" operand): a lane's B fragments are 8 consecutive channels of ITS pixel,
//     and after v_permlane32_swap its accumulators are too, so the fp16 output of one contraction
//     IS the B operand of the next - no LDS, no shuffles across waves. 96 registers hold t2, then
//     y1, then y; ffn.2's accumulators (192 registers) live across the whole walk over ffn.0.
//   * only weights move through LDS: one linear stream of 16 KB slabs ([128 out channels][64 k],
//     16 MFMAs per wave each) through a 5-slot ring fed by global_load_lds, four slabs in flight,
//     one barrier per slab. 1.77 MB of weights per workgroup from the XCD's L2 = 32 B/clk/CU at
//     full matrix-core rate. All waves of a CU share every slab: A-fragment reads are 128 B/clk/CU.
//   * contraction order, bias-initialised accumulators, epilogue order and rounding points are
//     conv_gemm.hip's, operation for operation: bit-identical to the launch sequence it replaces
//     (tests/test_kernels_gpu.py::test_dcb_core_equals_launch_sequence) and to the oracle.
#include "arith.h"
#include "ops.h"
#include "wsilu_table.h"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <type_traits>

namespace dcvc {

const float4* wsilu_table_device();      // conv_gemm.hip

namespace {

constexpr int C = 384;                   // block width
constexpr int NTHREADS = 256;
constexpr int BM = 128;                  // pixels per workgroup
constexpr int SLAB = 16384;              // one weight slab: [128 channels][64 k] or [64 channels][128 k] fp16
constexpr int NS = 5;                    // ring slots (the 15 slabs of a loop body map onto fixed slots)
constexpr int LPS = SLAB / (NTHREADS * 16);   // global_load_lds per thread and slab = 4
constexpr int R = 4;                     // interleaved copies of the WSiLU table
constexpr int TABLE_BYTES = WSILU_SEGMENTS * 16;
constexpr int OFF_TABLE = NS * SLAB;
// constants: the biases b3 | b0 | b2 | b1n as FP32 (an accumulator tile is initialised by four 16-byte LDS reads
// straight into the accumulator registers: as fp16 it took 16 conversions + 16 v_accvgpr_write per tile, 2.25 VALU
// operations per ffn.0 output element - a sixth of the walk's VALU work), then q | q2 as fp16
constexpr int CONST_HALVES = C + 4 * C + C + C + C + C;    // 16-byte units are fetched as fp16
constexpr int BIAS_FLOATS = 7 * C;
constexpr int OFF_CONST = OFF_TABLE + R * TABLE_BYTES;
constexpr int OFF_QSCALE = OFF_CONST + BIAS_FLOATS * 4;    // q | q2, fp16
constexpr int OFF_SLABS = OFF_QSCALE + 2 * C * 2;          // weight stream: address | shape of every slab, 8 B each
constexpr int SLAB_ENTRIES = 136;
constexpr int OFF_STAGE = OFF_SLABS + SLAB_ENTRIES * 8;
// per wave: [finished accumulator pair, 32 floats per lane, lane-linear 16-B units: 8 KB][64-channel output rows]
// the 128-channel output rows of the ffn.2 epilogue reuse the front of the area (nothing else is live then)
constexpr int DUMP_BYTES = 8 * 1024;
constexpr int PITCH128 = 256 + 16, PITCH64 = 128 + 16;   // staged output rows are padded, not swizzled
constexpr int WAVE_AREA = DUMP_BYTES + 32 * PITCH64;
static_assert(32 * PITCH128 <= WAVE_AREA, "ffn.2 output rows must fit the wave's area");
constexpr int SMEM_BYTES = OFF_STAGE + 4 * WAVE_AREA;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
// the weight stream (slab index g):
//   [0, 18)     dc.3, wide slabs, k-step outer / 128-channel chunk inner
//   [18, 30)    ffn.0 of super-chunk 0: 4 channel pairs x 3 deep slabs
//   [30, 105)   super-chunks 1..5, 15 slabs each: pair 0 | ffn.2 of the previous super-chunk (3 wide) | pairs 1..3
//   [105, 108)  ffn.2 of super-chunk 5
//   [108, 126)  dc.0 of the next block: 6 channel pairs x 3 deep slabs (optional)
constexpr int G_A = 18, G_B0 = 12, G_LOOP = 75, G_F5 = 3, G_D = 18;
constexpr int G_CORE = G_A + G_B0 + G_LOOP + G_F5;
static_assert(15 % NS == 0 && (G_A + G_B0) % NS == 0, "the loop body must start on slot 0 every time");
static_assert(G_CORE + G_D + NS <= SLAB_ENTRIES, "slab table too small");

#ifndef DCB_CORE_NO_FENCE
#define SLICE_FENCE() __builtin_amdgcn_sched_barrier(0)      // the schedule is hand-made: nothing crosses a k-slice
#else
#define SLICE_FENCE()
#endif
#ifndef DCB_CORE_NO_LOADS_FENCE
#define LOADS_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define LOADS_FENCE()
#endif
enum : int { WIDE = 0, DEEP = 1 };       // slab shapes: [128 rows][64 k] (4 tiles x 4 k-slices) / [64 rows][128 k] (2 x 8)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct CoreParams {
    const half_t* t2;     // depthwise output [M][ldt]
    const half_t* x;      // block input [M][ldx]: residual of dc.3 (and of ffn.2 when `shortcut`)
    const half_t* w3;  const half_t* b3;      // [C][C], folded bias [C]
    const half_t* w0;  const half_t* b0;      // [4C][C], [4C]
    const half_t* w2;  const half_t* b2;      // [C][C], [C]
    const half_t* q;   const half_t* q2;      // optional per-channel scales (fused / after rounding)
    const half_t* w1n; const half_t* b1n;     // optional: dc.0 of the next block
    const float4* wsilu;
    half_t* y;            // [M][ldy]
    half_t* t1n;          // [M][ldt1] (with w1n)
    int ldt, ldx, ldy, ldt1;
    int M, shortcut;
    long long* timeline;  // optional [workgroups][64] shader-clock stamps of wave 0 (tools/core_timeline.py)
};

// One LDS-DMA piece (64 lanes x 16 B, lane l lands at lds_dst + 16 l) issued behind the compiler's back.
// With the builtin, hipcc's wait-count pass puts s_waitcnt vmcnt(0) in front of the next LDS read that
// might alias the destination - it cannot tell the ring from the WSiLU table, so every table gather of
// the interleaved epilogue drained the whole prefetch (measured: 2700 cycles per 16 KB slab). The
// ordering that matters is enforced by hand: counted vmcnt + s_barrier at the top of every step.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

// the same with a wave-uniform base and a 32-bit byte offset per lane
__device__ __forceinline__ void lds_dma16_s(const void* sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

template <bool TIMELINE>
__global__ void __launch_bounds__(NTHREADS, 1)
dcb_core_kernel(const CoreParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31;
    const int hi = lane >> 5;
    const int m0 = blockIdx.x * BM + wave * 32;
    const int mc = min(m0 + px, p.M - 1);        // this lane's pixel, clamped for loads
    const int G = G_CORE + (p.w1n != nullptr ? G_D : 0);

    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<size_t>((lptr_t)smem));   // LDS byte address of the ring
    // ---- weight stream: slab g -> first element and shape. Every matrix has row stride C.
    auto slab_info = [&](int g, int& shape) -> const half_t* {
        if (g < G_A) {                       // dc.3: k-step outer, 128-channel chunk inner
            shape = WIDE;
            return p.w3 + static_cast<size_t>(g % 3) * (128 * C) + (g / 3) * 64;
        }
        g -= G_A;
        if (g < G_B0) {                      // ffn.0, super-chunk 0: pair g/3, k third g%3
            shape = DEEP;
            return p.w0 + static_cast<size_t>(g / 3) * (64 * C) + (g % 3) * 128;
        }
        g -= G_B0;
        if (g < G_LOOP) {
            const int sc = 1 + g / 15, r = g % 15;
            if (r >= 3 && r < 6) {           // ffn.2 of super-chunk sc-1: 128-channel chunk r-3
                shape = WIDE;
                return p.w2 + static_cast<size_t>(r - 3) * (128 * C) + (sc - 1) * 64;
            }
            const int pair = r < 3 ? 0 : 1 + (r - 6) / 3;
            const int k3 = r < 3 ? r : (r - 6) % 3;
            shape = DEEP;
            return p.w0 + static_cast<size_t>(sc * 256 + pair * 64) * C + k3 * 128;
        }
        g -= G_LOOP;
        if (g < G_F5) {
            shape = WIDE;
            return p.w2 + static_cast<size_t>(g) * (128 * C) + 5 * 64;
        }
        g -= G_F5;
        shape = DEEP;
        return p.w1n + static_cast<size_t>(g / 3) * (64 * C) + (g % 3) * 128;
    };
    // staging: the LDS image of a slab is lane-linear (16-B unit u = j*256 + tid), so the bank swizzle
    // sits on the SOURCE side. wide: row u>>3, chunk (u&7) ^ ((row>>1)&7); deep: row u>>4, chunk (u&15) ^ (row&15)
    const int toff_w = (tid >> 3) * C + ((tid & 7) ^ ((tid >> 4) & 7)) * 8;     // + j * 32 rows
    const int toff_d = (tid >> 4) * C + ((tid & 15) ^ ((tid >> 4) & 15)) * 8;   // + j * 16 rows
    // The stream is tabulated once (LDS, address | shape in bit 0) so that the steady state has no
    // control flow at all; entries behind the last slab repeat it: the prefetch of "slab g+4" and the
    // counted waits then stay uniform to the very end (a few KB of redundant loads, drained at exit).
    unsigned long long* ltbl = reinterpret_cast<unsigned long long*>(smem + OFF_SLABS);
    if (tid < SLAB_ENTRIES) {
        int shape = WIDE;
        const half_t* base = slab_info(min(tid, G - 1), shape);
        ltbl[tid] = reinterpret_cast<unsigned long long>(base) | static_cast<unsigned long long>(shape);
    }
    // scalar slab base + 32-bit lane offset (the saddr form of the load): per piece two scalar adds instead of a
    // 64-bit vector add per lane - 60 VALU operations per super-chunk of the walk
    struct Pending { const half_t* base; unsigned voff; int jstride; unsigned dst; };
    unsigned long long next_entry = 0;      // table entry of the slab the NEXT step prefetches: read one step early
    auto plan_slab = [&](int g, int slot) {
        const unsigned long long e = next_entry;
        next_entry = ltbl[g + 1];
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(e));
        const unsigned hi32 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(e >> 32));
        const bool deep = (lo & 1u) != 0;
        const half_t* base = reinterpret_cast<const half_t*>((static_cast<unsigned long long>(hi32) << 32) | (lo & ~1u));
        Pending q;
        q.base = base;
        q.voff = static_cast<unsigned>(deep ? toff_d : toff_w) * 2u;
        q.jstride = deep ? 16 * C : 32 * C;
        q.dst = lds_base + slot * SLAB + wave * 1024;
        return q;
    };
    // Ablation switches (tools/probes/core_bench.hip builds; the RESULTS ARE WRONG with any of them):
    //   DCB_EXP_NODMA  no LDS-DMA of weight slabs behind the first four   DCB_EXP_NOBAR  no barrier per slab
    //   DCB_EXP_NOEPI  the WSiLU pieces of the ffn walk do nothing        DCB_EXP_NOGATHER  polynomial on a constant row
    auto issue_part = [&](const Pending& q, int j) {
#ifndef DCB_EXP_NODMA
        lds_dma16_s(q.base + j * q.jstride, q.voff, q.dst + j * (NTHREADS * 16));
#endif
    };

    // ---- L2 warm-up. Every workgroup streams the SAME weights at the same time, and L2 starts cold
    // at every kernel boundary: the first touch of a line misses for all 32 CUs of an XCD at once and
    // they all sit out the Infinity-Cache round trip - with 48 KB in flight per CU that capped the
    // stream at ~6-13 B/clk/CU (measured: 2700 cycles per 16 KB slab). So each workgroup first touches
    // ITS share of the whole stream (line q belongs to the workgroup with (blockIdx / 8) % 32 == q % 32;
    // blockIdx % 8 is the XCD): 2 loads per thread, all misses in flight together, and the demand
    // loads behind them find their lines in L2.
    unsigned warm = 0;
    {
        const int rank = (blockIdx.x >> 3) & 31;
        constexpr int L3 = C * C / 64, L0 = 4 * C * C / 64;       // 128-B lines per matrix
        const int total = L3 + L0 + L3 + (p.w1n != nullptr ? L3 : 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int q = rank + 32 * (tid + NTHREADS * k);
            if (q < total) {
                const half_t* base = q < L3 ? p.w3 + static_cast<size_t>(q) * 64
                                   : q < L3 + L0 ? p.w0 + static_cast<size_t>(q - L3) * 64
                                   : q < 2 * L3 + L0 ? p.w2 + static_cast<size_t>(q - L3 - L0) * 64
                                   : p.w1n + static_cast<size_t>(q - 2 * L3 - L0) * 64;
                warm ^= *reinterpret_cast<const unsigned*>(base);
            }
        }
    }

    // ---- constants -> LDS (WSiLU table in R interleaved copies: lane l gathers from copy l & (R-1);
    // biases and scales). Their global loads go out FIRST: vmcnt retires in order, so a register-
    // destination load behind the weight prefetch drains the ring when it is waited for - for the
    // same reason no such load sits inside the slab loop except where the count is exact.
    constexpr int TAB_PER_THREAD = R * WSILU_SEGMENTS / NTHREADS;
    float4 tabv[TAB_PER_THREAD];
#pragma unroll
    for (int k = 0; k < TAB_PER_THREAD; ++k) tabv[k] = p.wsilu[(tid + k * NTHREADS) / R];
    constexpr int CONST_UNITS = CONST_HALVES / 8;        // 432 16-B units
    half8 constv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int ch = min(tid + k * NTHREADS, CONST_UNITS - 1) * 8;
        const half_t* src = ch < C ? p.b3 + ch
                          : ch < 5 * C ? p.b0 + (ch - C)
                          : ch < 6 * C ? p.b2 + (ch - 5 * C)
                          : ch < 7 * C ? (p.b1n != nullptr ? p.b1n + (ch - 6 * C) : p.b2)
                          : ch < 8 * C ? (p.q != nullptr ? p.q + (ch - 7 * C) : p.b2)
                          : (p.q2 != nullptr ? p.q2 + (ch - 8 * C) : p.b2);
        constv[k] = *reinterpret_cast<const half8*>(src);
    }

    // ---- B fragments of this lane's pixel: k-slice i = channels 16 i + 8 hi .. + 7
    half8 bf[24];
    {
        const half_t* row = p.t2 + static_cast<size_t>(mc) * p.ldt + 8 * hi;
#pragma unroll
        for (int i = 0; i < 24; ++i) bf[i] = *reinterpret_cast<const half8*>(row + 16 * i);
    }
#pragma unroll
    for (int g = 0; g < NS - 1; ++g) {        // slabs 0..3 are dc.3's (wide), straight from the matrix
        int shape = WIDE;
        Pending q;
        q.base = slab_info(g, shape);
        q.voff = static_cast<unsigned>(toff_w) * 2u;
        q.jstride = 32 * C;
        q.dst = lds_base + g * SLAB + wave * 1024;
#pragma unroll
        for (int j = 0; j < LPS; ++j) issue_part(q, j);
    }

    // LDS address of this lane's copy of the WSiLU table; its bits must stay clear of a row offset's (6..13) so
    // that index mask and base are one v_and_or_b32 (arith.h wsilu_row_lds): the table sits at a multiple of 16 KB
    static_assert(OFF_TABLE % 16384 == 0 && R == 4, "WSiLU table placement");
    const unsigned tab = lds_base + OFF_TABLE + (lane & (R - 1)) * 16;
    if ((lds_base & 16383u) != 0) __builtin_trap();      // dynamic LDS starts at 0 (no static LDS in this kernel)
    float* lbias = reinterpret_cast<float*>(smem + OFF_CONST);
    half_t* lqs = reinterpret_cast<half_t*>(smem + OFF_QSCALE);
    {
        float4* t = reinterpret_cast<float4*>(smem + OFF_TABLE);
#pragma unroll
        for (int k = 0; k < TAB_PER_THREAD; ++k) t[tid + k * NTHREADS] = tabv[k];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = tid + k * NTHREADS;
            if (u < BIAS_FLOATS / 8) {
                float4v lo4, hi4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo4[e] = static_cast<float>(constv[k][e]);
                    hi4[e] = static_cast<float>(constv[k][4 + e]);
                }
                *reinterpret_cast<float4v*>(lbias + u * 8) = lo4;
                *reinterpret_cast<float4v*>(lbias + u * 8 + 4) = hi4;
            } else if (u < CONST_UNITS) {
                *reinterpret_cast<half8*>(lqs + (u - BIAS_FLOATS / 8) * 8) = constv[k];
            }
        }
    }
    const float* lb3 = lbias;
    const float* lb0 = lbias + C;
    const float* lb2 = lbias + 5 * C;
    const float* lb1n = lbias + 6 * C;
    const half_t* lq = lqs;
    const half_t* lq2 = lqs + C;
    int stamp_no = 0;
    auto stamp = [&]() {
        if (TIMELINE && tid == 0 && stamp_no < 64) {
            p.timeline[static_cast<size_t>(blockIdx.x) * 64 + stamp_no] = static_cast<long long>(__builtin_readcyclecounter());
        }
        ++stamp_no;
    };

    // fine timeline (TIMELINE launches only): issue times of wave 0 inside two dc.3 steps (slabs 6 and 7) and two
    // ffn.2 steps of the walk - kept in scalar registers, written at the very end (slots 16.. of the row)
    unsigned fine_t[24] = {};
    auto fine = [&](int idx) {
        if constexpr (TIMELINE) fine_t[idx] = static_cast<unsigned>(__builtin_readcyclecounter());
    };
    int flush_probe = -1;            // >= 0: the next flush stamps its stages into fine_t[flush_probe ..]

    // A-fragment offsets inside a slab (tile t adds 32 rows)
    // One register per (shape, k-slice); slot and tile are compile-time byte offsets that fold into the
    // ds_read immediate. (Anything computed per (slot, tile, slice) is loop-invariant and gets hoisted:
    // 240 address registers in an earlier version of this kernel, i.e. spills all over the walk.)
    int foff_w[4], foff_d[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff_w[s] = (px * 128 + (((s * 2 + hi) ^ ((px >> 1) & 7)) << 4)) & 0xfff;
#pragma unroll
    for (int s = 0; s < 8; ++s) foff_d[s] = (px * 256 + (((s * 2 + hi) ^ (px & 15)) << 4)) & 0x1fff;
    // (the masks are no-ops that tell the compiler the bases are small and non-negative: without that
    // knowledge it will not fold the slot offset into the ds_read immediate)
    auto frag_w = [&](const char* ws, int tile, int s) { return *reinterpret_cast<const half8*>(ws + foff_w[s] + tile * 4096); };
    auto frag_d = [&](const char* ws, int tile, int s) { return *reinterpret_cast<const half8*>(ws + foff_d[s] + tile * 8192); };

    // slab 0 (and the constants) landed and visible; three more slabs stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp();                                         // 0: first slab there
    next_entry = ltbl[NS - 1];
    if (warm == 0x9e3779b9u && p.M < 0) p.y[0] = static_cast<half_t>(0);     // never true: keeps the warm-up loads alive
    // One slab = 16 MFMAs per wave. The barrier at the top of step g certifies slab g+1 (every wave
    // waited for its own share) and frees the slot of slab g-1 for the prefetch of slab g+4; slab g
    // itself was certified one step earlier, so nothing waits between the barrier and the first reads.
    // (Reading the first fragments of slab g+1 at the end of step g would hide one LDS round trip per
    // step, but the 16 registers that then live across every step boundary push hipcc into spilling
    // B fragments inside the walk - measured: 176 spilled registers with, 14 outside the loops without.)
    // piece(s): VALU work (the epilogue of an EARLIER accumulator set) placed behind the MFMAs of k-slice s.
    // `extra`: register-destination loads issued within the last three steps (they sit between the
    // stream's own operations in the in-order queue and must not be mistaken for them)
    // A NEGATIVE `extra` = that many global STORES (gfx950 counts stores in vmcnt too): a row store is skipped
    // when none of its rows exists, so only a wave whose 32 pixels are all inside the picture may count on them.
    const bool wave_full = m0 + 32 <= p.M;
#ifdef DCB_CORE_STAGED_STORES
    const bool wave_counts_stores = wave_full;
#else
    const bool wave_counts_stores = m0 < p.M;       // direct row stores: issued by every wave that owns a pixel
#endif
    auto step_top = [&](int g, int slot, auto extra) {
        constexpr int ex = decltype(extra)::value;
        if constexpr (ex >= 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS + ex) : "memory");
        } else {
            if (wave_counts_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS - ex) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef DCB_EXP_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        return plan_slab(g + NS - 1, (slot + NS - 1) % NS);
    };
    // A lone wave per SIMD issues in order: four MFMAs in a row block the wave for three pipe times
    // and everything else is issued afterwards with the matrix pipe idle (measured: 190 instead of
    // 128 cycles per 4-MFMA slice). The instruction groups below ask the scheduler for
    // MFMA | a few LDS reads | a few VALU | MFMA | ... inside each region (a region ends at the
    // LDS-DMA asm, which nothing crosses).
#ifdef DCB_CORE_SGB
#define SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SGB_DSRD(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define SGB_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
#else
#define SGB_MFMA(n)
#define SGB_DSRD(n)
#define SGB_VALU(n)
#endif
    // head[]: the first fragments of a step, read at the END of the step before (the slab was certified
    // by that step's barrier): tiles 0..3 of k-slice 0 (wide) / tiles 0,1 of k-slices 0,1 (deep).
    // next_tag: shape of the following slab, or NONE = the following step reads its own head
    // (have_head = false there); both are compile-time at every call site.
    half8 head[4];
    auto load_head = [&](auto shape_tag, const char* ws) {
        if constexpr (decltype(shape_tag)::value == WIDE) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) head[nt] = frag_w(ws, nt, 0);
        } else {
            head[0] = frag_d(ws, 0, 0); head[1] = frag_d(ws, 1, 0);
            head[2] = frag_d(ws, 0, 1); head[3] = frag_d(ws, 1, 1);
        }
    };
    auto step_wide = [&](int g, int slot, auto have_head, auto next_tag, auto&& bfrag, float16v (&acc)[4], auto&& pre, auto&& piece,
                         auto vtag, auto extra) {
        constexpr int valu_per_mfma = decltype(vtag)::value;
        const int fbase = (decltype(vtag)::value == 0 && g == 7) ? 0 : -1;     // vtag 0 = the dc.3 steps (fully unrolled: g is a constant there)
        if (TIMELINE && fbase >= 0) fine(fbase);
        const Pending nx = step_top(g, slot, extra);
        if (TIMELINE && fbase >= 0) fine(fbase + 1);
        const char* ws = smem + slot * SLAB;
        if constexpr (!decltype(have_head)::value) load_head(std::integral_constant<int, WIDE>{}, ws);
        half8 wf[2][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[0][nt] = head[nt];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[(s + 1) & 1][nt] = frag_w(ws, nt, s + 1);
            } else if constexpr (decltype(next_tag)::value >= 0) {
                load_head(next_tag, smem + ((slot + 1) % NS) * SLAB);
            }
            pre(s);
            // the reads of the NEXT slice's fragments go out in front of this slice's MFMAs: left alone, hipcc
            // sinks them behind three of the four MFMAs to recycle the fragment registers, and the next slice
            // then waits a full LDS round trip with the matrix pipe idle (measured in the ISA: lgkmcnt waits one
            // MFMA behind their reads)
            LOADS_FENCE();
            const half8 b = bfrag(s);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s & 1][nt], b, acc[nt], 0, 0, 0);
            piece(s);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                SGB_MFMA(1);
                SGB_DSRD(2);
                if constexpr (valu_per_mfma > 0) SGB_VALU(valu_per_mfma);
            }
            // (tried in round 3: every wave issuing its piece behind a different MFMA of the slice, so that the four
            // lockstep waves do not reach the address path together - 134 instead of 101 us: the scalar branches
            // cut the slice into four scheduling regions)
            issue_part(nx, s);
            SLICE_FENCE();
            if (TIMELINE && fbase >= 0) fine(fbase + 2 + s);
        }
    };
    auto step_deep = [&](int g, int slot, auto have_head, auto next_tag, auto&& bfrag, float16v (&acc)[2], auto&& pre, auto&& piece,
                         auto vtag, auto extra) {
        constexpr int valu_per_mfma = decltype(vtag)::value;
        // vtag 11 = the next block's dc.0; its pair 1 (steps G_CORE + 3 .. + 5) sits outside the loop: g is a constant
        const int fbase = (valu_per_mfma == 11 && g == G_CORE + 5) ? 18 : -1;
        if (TIMELINE && fbase >= 0) fine(fbase);
        const Pending nx = step_top(g, slot, extra);
        if (TIMELINE && fbase >= 0) fine(fbase + 1);
        const char* ws = smem + slot * SLAB;
        if constexpr (!decltype(have_head)::value) load_head(std::integral_constant<int, DEEP>{}, ws);
        half8 wf[3][2];
        wf[0][0] = head[0]; wf[0][1] = head[1]; wf[1][0] = head[2]; wf[1][1] = head[3];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 2 < 8) {
                wf[(s + 2) % 3][0] = frag_d(ws, 0, s + 2);
                wf[(s + 2) % 3][1] = frag_d(ws, 1, s + 2);
            } else if (s == 6) {
                if constexpr (decltype(next_tag)::value >= 0) load_head(next_tag, smem + ((slot + 1) % NS) * SLAB);
            }
            pre(s);
            LOADS_FENCE();
            const half8 b = bfrag(s);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % 3][0], b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % 3][1], b, acc[1], 0, 0, 0);
            piece(s);
            if (s & 1) {                 // one region = two k-slices = 4 MFMAs, closed by the LDS-DMA piece
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    SGB_MFMA(1);
                    SGB_DSRD(2);
                    if constexpr (valu_per_mfma > 0) SGB_VALU(valu_per_mfma);
                }
                issue_part(nx, s >> 1);
                SLICE_FENCE();
                if (TIMELINE && fbase >= 0) fine(fbase + 2 + (s >> 1));
            }
        }
    };
    using TagW = std::integral_constant<int, WIDE>;
    using TagD = std::integral_constant<int, DEEP>;
    using TagNone = std::integral_constant<int, -1>;
    using Yes = std::true_type;
    using No = std::false_type;
    auto no_piece = [](int) {};

    // accumulator tile (32 channels from `first`) initialised with the bias:
    // acc[r] = channel first + 8 (r>>2) + 4 hi + (r&3)
    auto bias_tile = [&](float16v& acc, const float* bias, int first) {
        const float* bp = bias + first + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4v b4 = *reinterpret_cast<const float4v*>(bp + 8 * g4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g4 + e] = b4[e];
        }
    };
    // accumulator tile -> the two 8-channel runs this lane owns after pairing the half-waves:
    // run pr = channels 16 pr + 8 hi .. + 7 of the tile, this lane's pixel (= B-fragment layout)
    auto runs_of = [&](const float16v& a, int pr, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * pr + e]),
                                                             __float_as_uint(a[8 * pr + 4 + e]), false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
        }
    };
    // NCH output channels (64 or 128) of the wave's 32 pixels -> memory in whole rows, via the wave's
    // own staging area (no other wave touches it: wave-local ordering is enough)
    char* const dump = smem + OFF_STAGE + wave * WAVE_AREA;
    auto flush = [&](const half8* o, auto nch_tag, half_t* dst, int ld, int first) {
        constexpr int NCH = decltype(nch_tag)::value;
        constexpr int CPR = NCH / 8;                       // 16-B chunks per row
        constexpr int STAGE_PITCH = NCH == 128 ? PITCH128 : PITCH64;
        char* const stg = NCH == 128 ? dump : dump + DUMP_BYTES;
        if (TIMELINE && flush_probe >= 0) fine(flush_probe);
#pragma unroll
        for (int i = 0; i < CPR / 2; ++i) {                // run i = (tile i/2, pr i%2) -> channels 16 i + 8 hi
            *reinterpret_cast<half8*>(stg + px * STAGE_PITCH + hi * 16 + i * 32) = o[i];
        }
        if (TIMELINE && flush_probe >= 0) fine(flush_probe + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (TIMELINE && flush_probe >= 0) fine(flush_probe + 2);
        constexpr int RPI = 64 / CPR;                      // rows per iteration
        const int rrow = lane / CPR, rc = lane % CPR;
        if (wave_full) {
            // all 32 rows exist: every read first, ONE wait, then the stores. (The guarded form below compiles to
            // a chain of branch | ds_read | s_waitcnt lgkmcnt(0) | store blocks: one exposed LDS round trip per
            // row group - measured 1 300 cycles per 64-channel flush, 24 such blocks in the y epilogue.)
            half8 v[CPR / 2];
#pragma unroll
            for (int it = 0; it < CPR / 2; ++it)
                v[it] = *reinterpret_cast<const half8*>(stg + rrow * STAGE_PITCH + rc * 16 + it * (RPI * STAGE_PITCH));
            if (TIMELINE && flush_probe >= 0) fine(flush_probe + 3);
            half_t* const row0 = dst + static_cast<size_t>(m0 + rrow) * ld + first + rc * 8;
            // (timing experiment, TIMELINE launches only: DCVC_CORE_NOSTORE=1 drops the stores)
            if (!(TIMELINE && (p.shortcut & 2))) {
#pragma unroll
                for (int it = 0; it < CPR / 2; ++it) store_line(row0 + static_cast<size_t>(it * RPI) * ld, v[it]);
            }
        } else {
#pragma unroll
            for (int it = 0; it < CPR / 2; ++it) {
                const int row = it * RPI + rrow;
                const half8 v = *reinterpret_cast<const half8*>(stg + rrow * STAGE_PITCH + rc * 16 + it * (RPI * STAGE_PITCH));
                if (m0 + row < p.M) {
                    store_line(dst + static_cast<size_t>(m0 + row) * ld + first + rc * 8, v);
                }
            }
        }
        if (TIMELINE && flush_probe >= 0) fine(flush_probe + 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (TIMELINE && flush_probe >= 0) fine(flush_probe + 5);
    };

    // The same rows WITHOUT the LDS round trip: every lane stores its own 16-byte runs (32 B per pixel row and
    // instruction, the four / eight runs of a row complete a 128-B line in L2). One instruction per run, issued by
    // every wave that owns at least one pixel.
    auto store_runs = [&](const half8* o, auto nch_tag, half_t* dst, int ld, int first) {
        constexpr int NCH = decltype(nch_tag)::value;
        if (m0 + px < p.M && !(TIMELINE && (p.shortcut & 2))) {     // (DCVC_CORE_NOSTORE: timing experiment)
            half_t* const row = dst + static_cast<size_t>(m0 + px) * ld + first + 8 * hi;
#pragma unroll
            for (int i = 0; i < NCH / 16; ++i) store_line(row + 16 * i, o[i]);
        }
    };
#ifdef DCB_CORE_STAGED_STORES
#define EMIT_ROWS flush
#else
#define EMIT_ROWS store_runs
#endif

    int g = 0;
    // ================================================================ dc.3: y1 = W3 t2 + b3' + x
    // All 12 channel tiles accumulate at once in the registers ffn.2's accumulators take over
    // afterwards; the fp16 result then replaces t2 in `bf` in place. The block input x arrives in the
    // registers t2 frees up (one 128-channel third after each of the k-steps 2..4): behind each of those loads the
    // number of younger memory operations is exact, so waiting for it does not drain the prefetch.
    float16v acc2[12];
#pragma unroll
    for (int nt = 0; nt < 12; ++nt) bias_tile(acc2[nt], lb3, 32 * nt);
    // The block input x (residual of dc.3) comes in three 128-channel thirds by LDS-DMA into the wave's own staging
    // area (free until the ffn walk parks accumulators there), whole 256-B row segments per instruction, bank
    // swizzle on the source side like the weight slabs, and is read from there in B-fragment layout into the
    // registers t2 frees up. (Round 2 first loaded it straight into registers, 32 B per row and lane pair: 32
    // partial lines per instruction - measured 1 800 instead of 780 cycles for every slab step behind such a load.)
    // Third t goes out behind step {2, 7, 13}: for the three steps that follow, its 8 pieces are younger than the
    // slab the barrier certifies (extra = 8), the fourth step certifies them; it is read behind step {6, 12, 17}
    // (the step in between has drained the reads of the previous third before the next one overwrites the area).
    half8 xr[24];
    const half_t* xrow = p.x + static_cast<size_t>(mc) * p.ldx + 8 * hi;
    const unsigned xdump = lds_base + OFF_STAGE + wave * WAVE_AREA;
    // scalar base + 32-bit lane offset: with 64-bit lane addresses hipcc keeps all of them (they do not depend
    // on the third) from the first third to the last and spills them - every reload is an s_waitcnt vmcnt(0),
    // i.e. a drained prefetch
    // (a wave entirely behind the last pixel reads the last row: the lane offset is unsigned)
    const int m0x = min(m0, p.M - 1);
    const half_t* const xwave = p.x + static_cast<size_t>(m0x) * p.ldx;     // wave-uniform
    const int xlast = p.M - 1 - m0x;                                         // last row of the picture, relative: >= 0
    auto x_dma = [&](int t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 4 * j + (lane >> 4);
            const int chunk = (lane & 15) ^ (row & 15);
            const unsigned off = static_cast<unsigned>(min(row, xlast) * p.ldx + 8 * chunk) * 2u;
            lds_dma16_s(xwave + 128 * t, off, xdump + j * 1024);
        }
    };
    auto x_read = [&](int t) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            xr[8 * t + i] = *reinterpret_cast<const half8*>(dump + px * 256 + (((2 * i + hi) ^ (px & 15)) << 4));
    };
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            auto run = [&](auto have, auto next, auto extra) {
                step_wide(g, (3 * ks + c) % NS, have, next, [&](int s) { return bf[ks * 4 + s]; },
                          *reinterpret_cast<float16v(*)[4]>(&acc2[4 * c]), no_piece, no_piece, std::integral_constant<int, 0>{}, extra);
            };
            using X0 = std::integral_constant<int, 0>;
            using X8 = std::integral_constant<int, 8>;
            const int st = 3 * ks + c;
            const bool x8 = (st >= 3 && st <= 5) || (st >= 8 && st <= 10) || (st >= 14 && st <= 16);
            if (st == 0) run(No{}, TagW{}, X0{});
            else if (st == 17) run(Yes{}, TagD{}, X0{});
            else if (x8) run(Yes{}, TagW{}, X8{});
            else run(Yes{}, TagW{}, X0{});
            ++g;
            if (st == 2) x_dma(0);
            if (st == 7) x_dma(1);
            if (st == 13) x_dma(2);
            if (st == 6) x_read(0);
            if (st == 12) x_read(1);
        }
    }
    x_read(2);
    stamp();                                         // 1: dc.3 slabs done
#pragma unroll
    for (int nt = 0; nt < 12; ++nt)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float v[8];
            runs_of(acc2[nt], pr, v);
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = to_half(v[e] + static_cast<float>(xr[2 * nt + pr][e]));
            // opaque: otherwise hipcc keeps every element a second time, unpacked, for the
            // residual add of the ffn.2 epilogue (192 values spilled across the ffn walk)
            asm volatile("" : "+v"(o));
            bf[2 * nt + pr] = o;
        }

    // ================================================================ ffn.0 -> t -> ffn.2 accumulators
    // ffn.0 runs channel pair by channel pair (2 tiles x the whole K = 3 deep slabs): the WSiLU /
    // chunk-add epilogue of a finished pair (8 pieces of 4 values + one combine -> ONE B fragment of
    // t) is issued between the MFMAs of the following 3 slabs, matrix pipe and VALU side by side.
#pragma unroll
    for (int nt = 0; nt < 12; ++nt) bias_tile(acc2[nt], lb2, 32 * nt);
    // A finished pair is parked in the wave's own LDS area (8 x 16-B units per lane, lane-linear) and
    // its epilogue reads 4 values at a time from there: reading them out of the accumulator registers
    // piecewise makes hipcc copy whole 16-register tuples to VGPRs for the duration (2 x 32 registers
    // across the walk = spills in the inner loop, which in-order vmcnt turns into full pipeline drains).
    float16v cur[2];
    float sums[2][4];
    auto park = [&](const float16v (&a)[2]) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4v v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = a[hh][4 * g4 + e];
                *reinterpret_cast<float4v*>(dump + (hh * 4 + g4) * 1024 + lane * 16) = v4;
            }
    };
    auto parked = [&](int unit) { return *reinterpret_cast<const float4v*>(dump + unit * 1024 + lane * 16); };
    half8 t3[4], t3n;
    // Epilogue of the parked pair, 8 pieces of 4 values + one combine. A piece is two DEPENDENT LDS round
    // trips (parked values -> table index -> coefficient gather -> polynomial); issued in one go behind the
    // MFMAs of a slice, the wave sat out both latencies with the matrix pipe idle after ~128 cycles
    // (measured: every interleaved piece cost its full ~300 cycles). In stages, each one slice apart from
    // the data it needs:  read (in front of the slice's MFMAs) | index + gather (behind them) | polynomial
    // + chunk sum (behind the MFMAs of the NEXT slice).
    float4v pv[2];                 // parked values of the pieces being read / indexed (q & 1)
    float ev[2][4];                // values of the pieces whose coefficients are in flight (q & 1)
    float4 ec[2][4];
    auto piece_read = [&](int q) { pv[q & 1] = parked(q); };
    // arithmetic policy v3 (arith.h): med3 | add | and-or -> table row; 2 fma + 1 fma (chunk sum) per element
    auto piece_index = [&](int q) {
#ifndef DCB_EXP_NOEPI
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ev[q & 1][e] = pv[q & 1][e];
#ifdef DCB_EXP_NOGATHER
            ec[q & 1][e] = make_float4(0.5f, 0.25f, 0.01f, 0.f);
#else
            ec[q & 1][e] = wsilu_row_lds<R, true>(pv[q & 1][e], tab);
#endif
        }
#endif
    };
    // (tried under policy v2: the operations with paired operands as packed fp32; hipcc pairs up only
    // half of them and pays for it in v_mov / s_nop: 1 538 instead of 1 400 VALU per super-chunk)
    auto piece_poly = [&](int q) {
#ifdef DCB_EXP_NOEPI
        sums[q >> 2][q & 3] = pv[q & 1][0];
        return;
#endif
        float acc = ev[q & 1][0] * wsilu_poly(ev[q & 1][0], ec[q & 1][0]);
#pragma unroll
        for (int e = 1; e < 4; ++e) acc = fmaf(ev[q & 1][e], wsilu_poly(ev[q & 1][e], ec[q & 1][e]), acc);
        sums[q >> 2][q & 3] = acc;
    };
    auto piece_combine = [&](half8& out) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sums[0][g4]),
                                                             __float_as_uint(sums[1][g4]), false, false);
            out[2 * g4] = to_half(__uint_as_float(sw[0]));
            out[2 * g4 + 1] = to_half(__uint_as_float(sw[1]));
        }
    };
    // 3 deep slabs of one ffn.0 channel pair (first channel `ch0`), with the epilogue of the previous
    // pair (-> out) spread over the 24 k-slices when `with_prev`
    auto ffn0_pair = [&](int slot0, int ch0, auto have_head, auto next_after, bool with_prev, half8& out) {
        bias_tile(cur[0], lb0, ch0);
        bias_tile(cur[1], lb0, ch0 + 32);
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3) {
            auto body = [&](auto have, auto next) {
                step_deep(g, (slot0 + k3) % NS, have, next, [&](int s) { return bf[8 * k3 + s]; }, cur,
                          [&](int s) {
                              const int slot24 = 8 * k3 + s;            // piece q: read in slice 3q, index in 3q+1, polynomial in 3q+2
                              if (with_prev && slot24 % 3 == 0) piece_read(slot24 / 3);
                          },
                          [&](int s) {
                              const int slot24 = 8 * k3 + s;
                              if (with_prev && slot24 % 3 == 1) piece_index(slot24 / 3);
                              if (with_prev && slot24 % 3 == 2) piece_poly(slot24 / 3);
                              if (with_prev && slot24 == 23) piece_combine(out);
                          }, std::integral_constant<int, 10>{}, std::integral_constant<int, 0>{});
            };
            if (k3 == 0) body(have_head, TagD{});
            else if (k3 == 1) body(Yes{}, TagD{});
            else body(Yes{}, next_after);
            ++g;
        }
        park(cur);
    };
    // 3 wide slabs of ffn.2 for the 64 channels of t in tt[0..3]; optionally carries an ffn.0 epilogue
    auto ffn2_group = [&](int slot0, const half8 (&tt)[4], auto have_head, auto next_after, bool with_prev, half8& out) {
#pragma unroll
        for (int nc = 0; nc < 3; ++nc) {
            auto body = [&](auto have, auto next) {
                step_wide(g, (slot0 + nc) % NS, have, next, [&](int s) { return tt[s]; },
                          *reinterpret_cast<float16v(*)[4]>(&acc2[4 * nc]),
                          [&](int s) {
                              const int slot12 = 4 * nc + s;            // piece q: read in slice q-1, index in q, polynomial in q+1
                              if (with_prev && slot12 == 0) piece_read(0);
                              if (with_prev && slot12 + 1 < 8) piece_read(slot12 + 1);
                          },
                          [&](int s) {
                              const int slot12 = 4 * nc + s;
                              if (with_prev && slot12 < 8) piece_index(slot12);        // gathers go out first ...
                              if (with_prev && slot12 >= 1 && slot12 <= 8) piece_poly(slot12 - 1);     // ... the previous piece's are there
                              if (with_prev && slot12 == 8) piece_combine(out);
                          }, std::integral_constant<int, 12>{}, std::integral_constant<int, 0>{});
            };
            if (nc == 0) body(have_head, TagW{});
            else if (nc == 1) body(Yes{}, TagW{});
            else body(Yes{}, next_after);
            ++g;
        }
    };
    auto finish_prev = [&](half8& out) {       // epilogue of the parked pair, not overlapped
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            piece_read(q);
            piece_index(q);
            piece_poly(q);
        }
        piece_combine(out);
    };

    stamp();                                         // 2: y1 epilogue done
    // super-chunk 0: pairs 0..3 (slots (18 + 3 j) % 5)
    ffn0_pair((G_A + 0) % NS, 0, Yes{}, TagD{}, false, t3[0]);
    ffn0_pair((G_A + 3) % NS, 64, Yes{}, TagD{}, true, t3[0]);
    ffn0_pair((G_A + 6) % NS, 128, Yes{}, TagD{}, true, t3[1]);
    ffn0_pair((G_A + 9) % NS, 192, Yes{}, TagNone{}, true, t3[2]);
    // super-chunks 1..5: pair 0 (finishes t3[3] of the previous one) | ffn.2 of the previous one (carries
    // pair 0's epilogue) | pairs 1..3. (The step behind the last pair differs between the iterations and
    // the exit, so the first step of an iteration reads its own head.)
    for (int sc = 1; sc < 6; ++sc) {
        stamp();                                     // 3..7: super-chunk sc starts
        ffn0_pair(0, sc * 256, No{}, TagW{}, true, t3[3]);
        ffn2_group(3, t3, Yes{}, TagD{}, true, t3n);
        t3[0] = t3n;
        ffn0_pair(6 % NS, sc * 256 + 64, Yes{}, TagD{}, false, t3n);
        ffn0_pair(9 % NS, sc * 256 + 128, Yes{}, TagD{}, true, t3[1]);
        ffn0_pair(12 % NS, sc * 256 + 192, Yes{}, TagNone{}, true, t3[2]);
    }
    stamp();                                         // 8: walk done
    finish_prev(t3[3]);
    ffn2_group(G_A + G_B0 + G_LOOP, t3, No{}, TagNone{}, false, t3n);
    stamp();                                         // 9: last ffn.2 group done

    // ================================================================ ffn.2 epilogue: y
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        half8 o8[8];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int i = 8 * c + 2 * nt + pr;           // k-slice index = channels 16 i + 8 hi ..
                float v[8];
                runs_of(acc2[4 * c + nt], pr, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(bf[i][e]);
                if (p.shortcut & 1) {
                    const half8 r8 = *reinterpret_cast<const half8*>(xrow + 16 * i);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(r8[e]);
                }
                if (p.q != nullptr) {
                    const half8 q8 = *reinterpret_cast<const half8*>(lq + 16 * i + 8 * hi);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * static_cast<float>(q8[e]);
                }
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
                if (p.q2 != nullptr) {
                    const half8 q8 = *reinterpret_cast<const half8*>(lq2 + 16 * i + 8 * hi);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = hmul(o[e], q8[e]);
                }
                bf[i] = o;
                o8[2 * nt + pr] = o;
            }
        EMIT_ROWS(o8, std::integral_constant<int, 128>{}, p.y, p.ldy, 128 * c);
    }

    stamp();                                         // 10: y written
    // ================================================================ dc.0 of the next block: t1' = WSiLU(W1' y + b1')
    if (p.w1n != nullptr) {
        half8 o4[4];
        // epilogue run r (tile r>>1, half r&1) of the parked pair, in the same three stages as the ffn.0 pieces
        // (read | index + gather | polynomial, one slice apart); dc0_flush: the pair's 64 channels go out
        // (4 values at a time: with all 8 table entries of a run in flight the 56 registers of state pushed address
        // registers of the slab walk into scratch, and every reload is an s_waitcnt vmcnt(0) = a drained prefetch)
        float4v dlo, dhi;
        float dv[8];
        float4 dc[4];
        auto dc0_read = [&](int r) {
            // run r = (tile r>>1, half r&1): parked units 2 (r&1) and 2 (r&1) + 1 of that tile, half-waves paired
            dlo = parked((r >> 1) * 4 + 2 * (r & 1));
            dhi = parked((r >> 1) * 4 + 2 * (r & 1) + 1);
        };
        auto dc0_pair_up = [&]() {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(dlo[e]), __float_as_uint(dhi[e]), false, false);
                dv[e] = __uint_as_float(sw[0]);
                dv[4 + e] = __uint_as_float(sw[1]);
            }
        };
        auto dc0_index = [&](int h) {          // table entries of values 4h .. 4h+3
#pragma unroll
            for (int e = 0; e < 4; ++e) dc[e] = wsilu_row_lds<R, true>(dv[4 * h + e], tab);
        };
        half8 orun;
        auto dc0_poly = [&](int r, int h) {
#pragma unroll
            for (int e = 0; e < 4; ++e) orun[4 * h + e] = to_half(dv[4 * h + e] * wsilu_poly(dv[4 * h + e], dc[e]));
            if (h == 1) o4[r] = orun;
        };
        auto dc0_flush = [&](int first) { EMIT_ROWS(o4, std::integral_constant<int, 64>{}, p.t1n, p.ldt1, first); };
        // `stores`: row stores (negative count, see step_top) younger than the slabs the first steps certify - the
        // flush of the previous pair sits behind 3 of the 4 prefetch pieces of the step before, the y epilogue
        // behind all of them
        auto dc0_pair = [&](int j, auto have_head, bool with_prev, auto st0, auto st1, auto st2) {
            bias_tile(cur[0], lb1n, 64 * j);
            bias_tile(cur[1], lb1n, 64 * j + 32);
#pragma unroll
            for (int k3 = 0; k3 < 3; ++k3) {
                auto body = [&](auto have, auto stores) {
                    step_deep(g, g % NS, have, TagD{}, [&](int s) { return bf[8 * k3 + s]; }, cur,
                              [&](int s) {
                                  const int slot24 = 8 * k3 + s;            // run r: read in slice 5r, then one stage per slice
                                  if (with_prev && slot24 % 5 == 0 && slot24 < 20) dc0_read(slot24 / 5);
                              },
                              [&](int s) {
                                  const int slot24 = 8 * k3 + s, r = slot24 / 5, ph = slot24 % 5;
                                  if (with_prev && slot24 < 20) {
                                      if (ph == 1) { dc0_pair_up(); dc0_index(0); }
                                      if (ph == 2) dc0_poly(r, 0);
                                      if (ph == 3) dc0_index(1);
                                      if (ph == 4) dc0_poly(r, 1);
                                  }
                                  // the 64 channels finished during the PREVIOUS pair go out right behind this step's
                                  // counted wait, when the fewest loads are in flight: with 11 LDS-DMA pieces outstanding
                                  // (slice 6 of the third step, where they used to sit) each row store took ~200 cycles
                                  // to issue, 830 per pair (measured with the stores compiled out)
                                  if (j >= 2 && slot24 == 0) dc0_flush(64 * (j - 2));
                              }, std::integral_constant<int, 11>{}, stores);
                };
                if (k3 == 0) body(have_head, st0);
                else if (k3 == 1) body(Yes{}, st1);
                else body(Yes{}, st2);
                ++g;
            }
            park(cur);
        };
        using S0 = std::integral_constant<int, 0>;
        using S4 = std::integral_constant<int, -4>;          // one 64-channel flush = 4 row stores per lane
        using S24 = std::integral_constant<int, -24>;        // the y epilogue = 3 x 8
        dc0_pair(0, No{}, false, S24{}, S24{}, S24{});
        dc0_pair(1, Yes{}, true, S0{}, S0{}, S0{});
        for (int j = 2; j < 6; ++j) dc0_pair(j, Yes{}, true, S0{}, S4{}, S4{});     // stores in slice 0 of the first step
        dc0_flush(64 * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dc0_read(r);
            dc0_pair_up();
            dc0_index(0);
            dc0_poly(r, 0);
            dc0_index(1);
            dc0_poly(r, 1);
        }
        dc0_flush(64 * 5);
    }
    // the prefetches behind the last slab are still on their way into this workgroup's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();                                         // 11: end
    if constexpr (TIMELINE) {
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 24; ++i) p.timeline[static_cast<size_t>(blockIdx.x) * 64 + 16 + i] = static_cast<long long>(fine_t[i]);
        }
    }
}

long long* g_core_timeline = nullptr;

}  // namespace

void dcb_core_timeline_buffer(long long* device_buffer)
{
    g_core_timeline = device_buffer;
}

bool dcb_core_supported(int c, int cdc, int cffn)
{
    static const bool off = [] { const char* e = getenv("DCVC_NO_DCB_CORE"); return e != nullptr && atoi(e) != 0; }();
    return !off && c == C && cdc == C && cffn == C;
}

void dcb_core(const DcbCoreDesc& d, hipStream_t stream)
{
    if (d.c != C) throw std::invalid_argument("dcb_core: block width must be 384");
    if (d.pixels <= 0) throw std::invalid_argument("dcb_core: empty problem");
    if ((d.ldt % 8) || (d.ldx % 8) || (d.ldy % 8) || (d.w1n && d.ldt1 % 8)) {
        throw std::invalid_argument("dcb_core: leading dimensions must be multiples of 8 channels");
    }
    if (!d.t2 || !d.x || !d.w3 || !d.b3 || !d.w0 || !d.b0 || !d.w2 || !d.b2 || !d.y || (d.w1n && (!d.b1n || !d.t1n))) {
        throw std::invalid_argument("dcb_core: missing operand");
    }
    CoreParams p{};
    p.t2 = d.t2; p.ldt = d.ldt; p.x = d.x; p.ldx = d.ldx;
    p.w3 = d.w3; p.b3 = d.b3; p.w0 = d.w0; p.b0 = d.b0; p.w2 = d.w2; p.b2 = d.b2;
    p.q = d.q; p.q2 = d.q2; p.w1n = d.w1n; p.b1n = d.b1n; p.t1n = d.t1n; p.ldt1 = d.ldt1;
    p.wsilu = wsilu_table_device();
    p.y = d.y; p.ldy = d.ldy; p.M = d.pixels; p.shortcut = d.shortcut ? 1 : 0;
    if (g_core_timeline != nullptr && getenv("DCVC_CORE_NOSTORE") != nullptr) p.shortcut |= 2;
    p.timeline = g_core_timeline;
    static std::once_flag once;
    std::call_once(once, [] {
        hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(dcb_core_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES),
                  "hipFuncSetAttribute(dcb_core)");
        hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(dcb_core_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES),
                  "hipFuncSetAttribute(dcb_core)");
    });
    hipEvent_t ev0, ev1;
    const int kflop = (d.w1n != nullptr ? 7 : 6) * C;            // 2 * pixels * C * kflop = FLOPs of the launch
    if (p.timeline == nullptr &&
        gemm_profile_slot(GemmLaunchInfo{d.pixels, C, kflop, static_cast<int>(0x80000000u), 0.f}, &ev0, &ev1)) {
        hipExtLaunchKernelGGL(dcb_core_kernel<false>, dim3((d.pixels + BM - 1) / BM), dim3(NTHREADS), SMEM_BYTES, stream,
                              ev0, ev1, 0, p);
    } else if (p.timeline != nullptr) {
        hipLaunchKernelGGL(dcb_core_kernel<true>, dim3((d.pixels + BM - 1) / BM), dim3(NTHREADS), SMEM_BYTES, stream, p);
    } else {
        hipLaunchKernelGGL(dcb_core_kernel<false>, dim3((d.pixels + BM - 1) / BM), dim3(NTHREADS), SMEM_BYTES, stream, p);
    }
    hip_check(hipGetLastError(), "dcb_core launch");
}

}  // namespace dcvc
