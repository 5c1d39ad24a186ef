// dcb_pair8_kernel.h - the two 1x1 convs in FRONT of a DepthConvBlock's depthwise conv in ONE launch (round 6):
//
//     in = Wa * x + ba                 the block's adaptor (layers.py:137-139: blocks whose input width differs from their own)
//     t1 = WSiLU(W1 * in + b1)         dc.0
//
// Reference: DepthConvBlockProxy::forward, layers_proxy.cpp:73-79 (conv1x1_bias, then conv1x1_bias_wsilu: two launches with
// `in` going through memory in between). Same arithmetic as conv_gemm / dcb_nsplit8 (bias-initialised accumulators, k
// ascending in slices of 16, one rounding to fp16 per tensor): bit-identical to conv1x1 + conv1x1(wsilu).
//
// Built like the block kernel (dcb_nsplit8_kernel.h), whose pieces it reuses: a workgroup of eight waves owns 32 * PXT
// pixels; their input rows (ALL CIN channels: CIN <= 512) arrive in LDS by LDS-DMA (A, swizzled), every wave owns an eighth
// of the adaptor's output channels and streams ITS weight fragments from a packed stream (dcb_pair_pack_adaptor) into a
// register ring, `in` goes to LDS (B: dc.0's operand) and to memory, dc.0 reads B with the block's own packed dc.0 stream
// (dcb_nsplit_pack_dc0: the same bytes the previous block's NEXT slot would use) and writes t1. Persistent workgroups.
#pragma once
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace pair8 {

using nsplit::lds_dma16;
using nsplit::static_for;
using nsplit::TABLE_BYTES;
using nsplit8::Geo;
using nsplit8::NTHREADS;
using nsplit8::RING;

struct PairParams {
    const half_t* x; int ldx;        // [M][ldx], first CIN channels
    const half8* wa;                 // packed adaptor weights (dcb_pair_pack_adaptor)
    const half8* w1;                 // packed dc.0 weights (dcb_nsplit_pack_dc0 of the block)
    const half_t* ba; const half_t* b1;
    const float4* wsilu;
    half_t* y; int ldy;              // `in` [M][ldy], C channels
    half_t* t1; int ldt1;            // dc.0 output [M][ldt1], CI channels
    int M;
};

template <int CIN, int C, int CI, int PXT>
struct Lay {
    static constexpr int CINP = (CIN + 127) / 128 * 128;              // LDS rows in groups of 16 chunks
    static constexpr int CP = (C + 127) / 128 * 128;                  // (block width 192: the intra decoder's last block)
    static constexpr int BUF_A = 32 * PXT * CINP * 2, BUF_B = 32 * PXT * CP * 2;
    static constexpr int RT = 4;
    static constexpr int OFF_B = BUF_A;
    static constexpr int OFF_TABLE = nsplit::align16k(BUF_A + BUF_B);
    static constexpr int OFF_BIAS = OFF_TABLE + RT * TABLE_BYTES;
    static constexpr int BYTES = OFF_BIAS + (CP + CI) * 4;
    static constexpr bool FITS = BYTES <= 160 * 1024;
};

template <int CIN, int C, int CI, int PXT, bool HIW>
__device__ __forceinline__ void pair_body(const PairParams& p, char* const smem)
{
    using G = Geo<C, CI>;
    using L = Lay<CIN, C, CI, PXT>;
    constexpr int PX = 32 * PXT;
    constexpr int KS_A = CIN / 16, KS_C = C / 16;
    constexpr int NT_C = G::nt_c(HIW), NT_N = G::nt_i(HIW);
    constexpr int F_A = NT_C * KS_A, F_1 = NT_N * KS_C, TOTAL = F_A + F_1;
    constexpr int CH_A = L::CINP / 8, PITCH_A = L::CINP * 2, PITCH_C = L::CP * 2;
    constexpr int RT = L::RT;
    constexpr int OFF_B = L::OFF_B, OFF_TABLE = L::OFF_TABLE, OFF_BIAS = L::OFF_BIAS;
    static_assert(CIN % 64 == 0 && (PX * CH_A) % NTHREADS == 0, "input rows must split evenly over the threads");
    static_assert(G::I_BY_PAIR || G::fin_ok(CI), "dc.0: an inner width the eight waves can share");
    static_assert(G::C_BY_PAIR || G::fin_ok(C), "a block width the eight waves can share");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int simd = wave & 3;
    const int px = lane & 31;
    const int hi = lane >> 5;
    const int ntiles = (p.M + PX - 1) / PX;
    int tile = blockIdx.x;
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) void*)smem));
    if ((lds_base & 16383u) != 0) __builtin_trap();
    const bool upper = wave >= 4;
    const int cb_c = G::C_BY_PAIR ? 32 * (simd * G::QC + (upper ? G::HI_C : 0))
                   : HIW ? 32 * wave * G::nf_hi(C) : 32 * (4 * G::nf_hi(C) + (wave - 4) * G::nf_lo(C));
    const bool c_on = G::C_BY_PAIR || HIW || (wave - 4) < G::act_lo(C);     // (block width 192: waves 6, 7 walk a tile of zeros)
    constexpr bool BY_PAIR = G::I_BY_PAIR;
    const int cb_n = BY_PAIR ? 32 * (simd * G::QI + (upper ? G::HI_I : 0))
                   : HIW ? 32 * wave * G::nf_hi(CI) : 32 * (4 * G::nf_hi(CI) + (wave - 4) * G::nf_lo(CI));
    const bool nx_on = BY_PAIR || HIW || (wave - 4) < G::act_lo(CI);

    // ---- constants: WSiLU table (RT interleaved copies), fp32 bias rows ba (C) | b1 (CI)
    {
        float4* t = reinterpret_cast<float4*>(smem + OFF_TABLE);
        t[tid] = p.wsilu[tid / RT];
        t[tid + NTHREADS] = p.wsilu[(tid + NTHREADS) / RT];
        float* lb = reinterpret_cast<float*>(smem + OFF_BIAS);
        for (int i = tid; i < L::CP + CI; i += NTHREADS) lb[i] = i < C ? static_cast<float>(p.ba[i]) : i < L::CP ? 0.f : static_cast<float>(p.b1[i - L::CP]);
    }
    const float* const lba = reinterpret_cast<const float*>(smem + OFF_BIAS);
    const float* const lb1 = lba + L::CP;
    unsigned tab = lds_base + OFF_TABLE + (lane & (RT - 1)) * 16;

    // ---- weight streams: waves 0 .. 3 first, then 4 .. 7 (their shares may differ); fragment f of a wave at 1 KB f
    constexpr int FA_HI = G::nt_c(true) * KS_A, FA_LO = G::nt_c(false) * KS_A, F1_HI = G::f_dc0(true), F1_LO = G::f_dc0(false);
    unsigned wsa = static_cast<unsigned>((HIW ? wave * FA_HI : 4 * FA_HI + (wave - 4) * FA_LO) * 64 + lane) * 16u;
    unsigned ws1 = static_cast<unsigned>((HIW ? wave * F1_HI : nx_on ? 4 * F1_HI + (wave - 4) * F1_LO : 0) * 64 + lane) * 16u;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8*>(p.wa), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8*>(p.w1), 0, 0x7fffffff, 0x00020000);
    half8 ring[RING];
    auto issue = [&](auto f_tag) {
        constexpr int f = decltype(f_tag)::value;
        if constexpr (f < F_A) {
            ring[f % RING] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, wsa + static_cast<unsigned>(f & 3) * 1024u, (f >> 2) * 4096, 0));
        } else if constexpr (f < TOTAL) {
            constexpr int g = f - F_A;
            ring[f % RING] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_1, ws1 + static_cast<unsigned>(g & 3) * 1024u, (g >> 2) * 4096, 0));
        }
    };
    // ---- the input rows of a tile -> A (lane-linear LDS image, the swizzle - chunk c of row r at c ^ (r & 15) inside its
    // group of 16 - on the source side; slots of the padding read any chunk of the row; rows behind the picture its last row)
    int tidv = tid;
    auto dma_tile = [&](int first_row) {
        const int f = min(first_row, p.M - 1);
        const int last = p.M - 1 - f;
        const half_t* const w = p.x + static_cast<size_t>(f) * p.ldx;
#pragma unroll
        for (int i = 0; i < PX * CH_A / NTHREADS; ++i) {
            const int pos = i * NTHREADS + tidv;
            const int r = pos / CH_A, pc = pos % CH_A;
            int lc = pc ^ (r & 15);
            if constexpr (CH_A * 8 != CIN) lc = min(lc, CIN / 8 - 1);
            const int rr = min(r, last);
            lds_dma16(w, static_cast<unsigned>(rr * p.ldx + lc * 8) * 2u, lds_base + (i * NTHREADS + wave * 64) * 16);
        }
    };
    int s0 = (hi ^ (px & 15)) << 4;
    int rowA = px * PITCH_A, rowB = px * PITCH_C + OFF_B;
    int hi4 = 4 * hi;
    int pxv = px, hiv = hi;
    int fa8[8], fb8[8];
    auto frag_a = [&](int t, int ks) { return *reinterpret_cast<const half8*>(smem + fa8[ks & 7] + (t * (32 * PITCH_A) + (ks >> 3) * 256)); };
    auto frag_b = [&](int t, int ks) { return *reinterpret_cast<const half8*>(smem + fb8[ks & 7] + (t * (32 * PITCH_C) + (ks >> 3) * 256)); };
    auto run_b = [&](int t, int ch0) { return reinterpret_cast<half8*>(smem + rowB + t * (32 * PITCH_C) + ((ch0 * 2) ^ s0)); };
    auto bias_tile = [&](float16v& acc, const float* bias, int first) {
        const float* bp = bias + first + hi4;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4v b4 = *reinterpret_cast<const float4v*>(bp + 8 * g4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g4 + e] = b4[e];
        }
    };
    auto runs_of = [&](const float16v& a, int pr, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * pr + e]), __float_as_uint(a[8 * pr + 4 + e]), false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
        }
    };
    auto contract = [&](auto nt_tag, auto ks_tag, auto f0_tag, auto&& frag, auto& acc) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr int KSN = decltype(ks_tag)::value;
        constexpr int F0 = decltype(f0_tag)::value;
        half8 b[2][PXT];
#pragma unroll
        for (int t = 0; t < PXT; ++t) b[0][t] = frag(t, 0);
        static_for<0, KSN>([&](auto kt) {
            constexpr int ks = decltype(kt)::value;
            if constexpr (ks + 1 < KSN) {
#pragma unroll
                for (int t = 0; t < PXT; ++t) b[(ks + 1) & 1][t] = frag(t, ks + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NT>([&](auto j_tag) {
                constexpr int j = NT - 1 - decltype(j_tag)::value;
                const half8 a = ring[(F0 + ks * NT + j) % RING];
#pragma unroll
                for (int t = 0; t < PXT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[ks & 1][t], acc[j][t], 0, 0, 0);
            });
            static_for<0, NT>([&](auto j_tag) { issue(std::integral_constant<int, F0 + ks * NT + decltype(j_tag)::value + RING>{}); });
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    dma_tile(tile * PX);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, RING>([&](auto i) { issue(i); });
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING) : "memory");       // this wave's pieces of the tile have landed
    __syncthreads();                                                   // ... everybody's, and the constants

    for (;;) {
        asm volatile("" : "+v"(wsa), "+v"(ws1), "+v"(tab), "+v"(s0), "+v"(rowA), "+v"(rowB), "+v"(hi4), "+v"(tidv), "+v"(pxv), "+v"(hiv));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa8[i] = rowA + ((32 * i) ^ s0);
            fb8[i] = rowB + ((32 * i) ^ s0);
        }
        const int m0 = tile * PX;
        const int next_tile = tile + static_cast<int>(gridDim.x);
        const bool has_next = next_tile < ntiles;
        // ============================================================ adaptor: in = Wa x + ba   (A -> B, memory)
        if constexpr (NT_C > 0) {
            float16v acc[NT_C][PXT];
#pragma unroll
            for (int j = 0; j < NT_C; ++j)
#pragma unroll
                for (int t = 0; t < PXT; ++t) bias_tile(acc[j][t], lba, cb_c + 32 * j);
            contract(std::integral_constant<int, NT_C>{}, std::integral_constant<int, KS_A>{}, std::integral_constant<int, 0>{}, frag_a, acc);
#pragma unroll
            for (int j = 0; j < NT_C; ++j)
#pragma unroll
                for (int t = 0; t < PXT; ++t)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float v[8];
                        runs_of(acc[j][t], pr, v);
                        half8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
                        const int ch = cb_c + 32 * j + 16 * pr;
                        *run_b(t, ch) = o;
                        const int m = m0 + 32 * t + pxv;
                        if (m < p.M && c_on) store_line(p.y + static_cast<size_t>(m) * p.ldy + ch + 8 * hiv, o);
                    }
        }
        __syncthreads();            // `in` complete in B; every wave is done with the input rows in A
        // the next tile's input rows (A is free) and the first weight fragments of its adaptor go out behind dc.0's MFMAs, in
        // front of an epilogue that needs no weights (a transfer in front of a contraction holds up every fragment requested
        // behind it: loads return in order)
        auto refill = [&]() {
            if (has_next) {
                dma_tile(next_tile * PX);
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, RING>([&](auto i) { issue(i); });
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ============================================================ dc.0: t1 = WSiLU(W1 in + b1)   (B -> memory)
        if constexpr (NT_N > 0) {
            if (nx_on) {
                float16v acc[NT_N][PXT];
#pragma unroll
                for (int j = 0; j < NT_N; ++j)
#pragma unroll
                    for (int t = 0; t < PXT; ++t) bias_tile(acc[j][t], lb1, cb_n + 32 * j);
                contract(std::integral_constant<int, NT_N>{}, std::integral_constant<int, KS_C>{}, std::integral_constant<int, F_A>{}, frag_b, acc);
                refill();
#pragma unroll
                for (int j = 0; j < NT_N; ++j)
#pragma unroll
                    for (int t = 0; t < PXT; ++t)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            float v[8];
                            runs_of(acc[j][t], pr, v);
                            float4v c[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float4 r = wsilu_row_lds<RT, true>(v[e], tab);
                                c[e] = float4v{r.x, r.y, r.z, r.w};
                            }
                            half8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = to_half(v[e] * wsilu_poly(v[e], make_float4(c[e][0], c[e][1], c[e][2], c[e][3])));
                            const int m = m0 + 32 * t + pxv;
                            if (m < p.M) store_line(p.t1 + static_cast<size_t>(m) * p.ldt1 + cb_n + 32 * j + 16 * pr + 8 * hiv, o);
                        }
            } else {
                refill();
            }
        } else {
            refill();
        }
        if (!has_next) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next tile's rows (and the first fragments) have landed
        __syncthreads();                                        // ... everybody's; every wave is done with B
        tile = next_tile;
    }
}

template <int CIN, int C, int CI, int PXT>
__global__ void __launch_bounds__(NTHREADS, 2)
dcb_pair8_kernel(const PairParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_pair[];
    if constexpr (Geo<C, CI>::EVEN) {
        pair_body<CIN, C, CI, PXT, true>(p, smem_pair);
    } else {
        if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) pair_body<CIN, C, CI, PXT, true>(p, smem_pair);
        else pair_body<CIN, C, CI, PXT, false>(p, smem_pair);
    }
}

template <int CIN, int C, int CI, int PXT>
void launch_pair(const PairParams& p, hipStream_t stream)
{
    auto kern = dcb_pair8_kernel<CIN, C, CI, PXT>;
    constexpr int smem = Lay<CIN, C, CI, PXT>::BYTES;
    static_assert(Lay<CIN, C, CI, PXT>::FITS, "LDS budget");
    constexpr int MAX_DEVICES = 64;
    static std::once_flag once[MAX_DEVICES];
    static int cu_count[MAX_DEVICES];
    int dev = 0;
    hip_check(hipGetDevice(&dev), "hipGetDevice");
    if (dev < 0 || dev >= MAX_DEVICES) throw std::runtime_error("dcb_pair8: device id out of range");
    std::call_once(once[dev], [&] {
        hipDeviceProp_t prop;
        hip_check(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
            throw std::runtime_error(std::string("dcb_pair8 needs gfx950 (160 KB LDS, permlane32_swap); device is ") + prop.gcnArchName);
        }
        hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem),
                  "hipFuncSetAttribute(dcb_pair8)");
        int n = 0;
        hip_check(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute");
        cu_count[dev] = n > 0 ? n : 256;
    });
    const int cus = cu_count[dev];
    const int tiles = (p.M + 32 * PXT - 1) / (32 * PXT);
    const int grid = tiles < cus ? tiles : cus;
    hipEvent_t ev0, ev1;
    // bench.py's roofline pass: 2 M C kflop = the launch's FLOPs (family 6 in bits 28..31, ops.h)
    const int kflop = CIN + CI;
    if (gemm_profile_slot(GemmLaunchInfo{p.M, C, kflop, 0x60000000, 0.f}, &ev0, &ev1)) {
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), smem, stream, ev0, ev1, 0, p);
    } else {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), smem, stream, p);
    }
    hip_check(hipGetLastError(), "dcb_pair8 launch");
}

template <int CIN, int C, int CI>
void run_pair(const PairParams& p, bool wide, hipStream_t stream)
{
    if constexpr (Lay<CIN, C, CI, 2>::FITS) {
        if (wide) { launch_pair<CIN, C, CI, 2>(p, stream); return; }
    }
    launch_pair<CIN, C, CI, 1>(p, stream);
}

}  // namespace pair8
}  // namespace dcvc
