// gemm_pipe.hip - EXPERIMENT, default OFF (DCVC_GEMM_PIPE=1), written at the end of round 1 after
// the GPU budget was spent: compiled and inspected (ISA, registers) but NOT YET RUN on hardware.
//
// conv1x1 + bias + WSiLU + chunk-add (ffn.0 of every DepthConvBlock, conv1x1_bias_wsilu_chunk_add.cu:
// 356-390) as a software-pipelined kernel: a workgroup owns 128 pixels and walks over all 256-channel
// tiles of N; the WSiLU epilogue of tile i runs INSIDE the main loop of tile i+1, interleaved with
// its MFMAs. Why: per SIMD and tile the matrix cores and the epilogue's VALU issue
// need about the same number of cycles, conv_gemm.hip runs them one after the other (plus an exposed
// prologue per tile), and two co-resident workgroups do not interleave them either.
//
//   * 4 waves (one per SIMD, up to 512 registers each), tile 128 x 256 x 64, each wave 64 x 128:
//     2 x 4 MFMA tiles, TWO accumulator sets of 128 registers (tile i being finished, tile i+1
//     being accumulated).
//   * the arithmetic is conv_gemm.hip's, operation for operation (accumulators start at the bias,
//     k ascends in 16-slices, WSiLU table, ((z0+z1)+z2)+z3, one rounding to fp16): results are
//     bit-identical, tests/test_kernels_gpu.py::test_gemm_pipe_equals_conv_gemm.
//   * LDS: 2 stages x (16 KB activations + 32 KB weights), 8 interleaved copies of the WSiLU table
//     (32 KB, bank-conflict-free gathers), the 16 KB output tile (whole 128-B lines to memory), the
//     bias vector: 148 KB + N * 2 B.
//   * the epilogue of a tile is cut into 16 pieces of 8 elements + 4 combine steps + 1 store step,
//     placed at fixed k-slices of the next tile (table below); a piece's table gathers are issued
//     one step before they are used.
#include "arith.h"
#include "ops.h"
#include "wsilu_table.h"

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <type_traits>
#include <utility>

namespace dcvc {

const float4* wsilu_table_device();      // conv_gemm.hip

namespace {

constexpr int BK = 64, BM = 128, BN = 256, NTHREADS = 256;
constexpr int WN = 2, MT = 2, NT = 4;                       // 2 x 2 waves, 64 x 128 per wave
constexpr int XT_BYTES = BM * BK * 2, WT_BYTES = BN * BK * 2, STAGE_BYTES = XT_BYTES + WT_BYTES;
constexpr int R = 8;                                         // interleaved table copies
constexpr int TABLE_BYTES = R * WSILU_SEGMENTS * 16;
constexpr int BNO = BN / 4;                                  // output channels per tile row
constexpr int OT_BYTES = BM * BNO * 2;
constexpr int OFF_TABLE = 2 * STAGE_BYTES, OFF_OTILE = OFF_TABLE + TABLE_BYTES, OFF_BIAS = OFF_OTILE + OT_BYTES;
constexpr int MAX_N = 4096;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct PipeParams {
    const half_t* x;
    const half_t* w;       // [N][K]
    const half_t* bias;    // [N]
    const float4* wsilu;
    half_t* y;             // [M][ldy], N/4 channels
    int ldx, ldy, M, N;
};

// ---- where the epilogue of the previous tile sits in the 4*NK k-slices of the current one
template <int NK> constexpr int piece_at(int q);      // piece index 0..15 or -1
template <int NK> constexpr int combine_at(int q);    // unit index 0..3 or -1
template <int NK> constexpr bool store_at(int q);     // the tile before the previous one goes to memory
template <> constexpr int piece_at<6>(int q) { return q % 3 != 2 ? (q / 3) * 2 + q % 3 : -1; }
template <> constexpr int combine_at<6>(int q) { return q % 6 == 5 ? q / 6 : -1; }
template <> constexpr bool store_at<6>(int q) { return q == 2; }
template <> constexpr int piece_at<8>(int q) { return q % 2 == 0 ? q / 2 : -1; }
template <> constexpr int combine_at<8>(int q) { return q % 8 == 7 ? q / 8 : -1; }
template <> constexpr bool store_at<8>(int q) { return q == 1; }

// every piece exactly once and in order, a unit's combine after its four pieces, the store before the
// first combine and behind the k-step barrier that follows the previous tile's last combine
template <int NK> constexpr bool schedule_ok()
{
    int next_piece = 0, next_unit = 0;
    bool stored = false;
    for (int q = 0; q < 4 * NK; ++q) {
        if (piece_at<NK>(q) >= 0) {
            if (piece_at<NK>(q) != next_piece || combine_at<NK>(q) >= 0) return false;
            ++next_piece;
        }
        if (combine_at<NK>(q) >= 0) {
            if (combine_at<NK>(q) != next_unit || next_piece != 4 * (next_unit + 1) || !stored || q < 4) return false;
            ++next_unit;
        }
        if (store_at<NK>(q)) {
            if (stored || next_unit != 0 || q >= 4) return false;
            stored = true;
        }
    }
    return next_piece == 16 && next_unit == 4 && stored;
}
static_assert(schedule_ok<6>() && schedule_ok<8>(), "epilogue schedule");

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int NK, bool SCHED>
__global__ void __launch_bounds__(NTHREADS, 1)
gemm_pipe_kernel(const PipeParams p)
{
    constexpr int K = NK * BK;
    static_assert(NK % 2 == 0, "stage buffer parity must not depend on the tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int hi = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int tiles_n = p.N / BN;

    // ---- staging plan (as conv_gemm.hip): 16-B unit u = j*256 + tid -> row u>>3, physical chunk
    //      u&7, logical chunk = physical ^ ((row>>1)&7)
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
    const half_t* xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xsrc[j] = p.x + static_cast<size_t>(min(m0 + j * 32 + srow, p.M - 1)) * p.ldx + schunk * 8;
    const half_t* wbase = p.w + static_cast<size_t>(srow) * K + schunk * 8;
    // k-slice `part` of a step carries a quarter of the next step's loads: X rows 32*part.., W rows 64*part..
    auto stage = [&](int buf, int n_i, int k0, int part) {
        char* xs = smem + buf * STAGE_BYTES;
        char* ws = xs + XT_BYTES;
        __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[part] + k0), (lptr_t)(xs + (part * NTHREADS + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int jj = 2 * part; jj < 2 * part + 2; ++jj) {
            const half_t* src = wbase + static_cast<size_t>(n_i * BN + jj * 32) * K + k0;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ws + (jj * NTHREADS + wave * 64) * 16), 16, 0, 0);
        }
    };

    // the first step goes out before anything else touches the memory pipeline
#pragma unroll
    for (int part = 0; part < 4; ++part) stage(0, 0, 0, part);

    // ---- one-time LDS tables: interleaved WSiLU copies (lane l reads copy l & 7), bias vector
    {
        float4* rep = reinterpret_cast<float4*>(smem + OFF_TABLE);
        for (int i = tid; i < R * WSILU_SEGMENTS; i += NTHREADS) rep[i] = p.wsilu[i / R];
        half4* bl = reinterpret_cast<half4*>(smem + OFF_BIAS);      // 8-B loads: conv_gemm.hip's alignment contract
        for (int i = tid; i < p.N / 4; i += NTHREADS) bl[i] = *reinterpret_cast<const half4*>(p.bias + 4 * i);
    }
    const float4* tab = reinterpret_cast<const float4*>(smem + OFF_TABLE) + (lane & (R - 1));
    char* otile = smem + OFF_OTILE;
    auto oaddr = [&](int row, int cidx) { return otile + row * (BNO * 2) + ((cidx ^ (row & 7)) << 4); };

    // ---- fragment read offsets (bytes inside a 32-row block) for the four 16-wide k slices
    const int frow = lane & 31;
    const int fsw = (frow >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = frow * 128 + (((s * 2 + hi) ^ fsw) << 4);

    float16v acc[2][NT][MT];
    // acc[.][nt][mt][r] belongs to channel n0 + (wn*NT + nt)*32 + 8*(r>>2) + 4*hi + (r&3); it starts
    // at the bias (arithmetic policy: y = ((bias + p_0) + p_1) + ...)
    auto init_acc = [&](auto set_c, int n_i) {
        constexpr int SET = decltype(set_c)::value;
        const half_t* bl = reinterpret_cast<const half_t*>(smem + OFF_BIAS) + n_i * BN + wn * (NT * 32) + 4 * hi;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            float16v init;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const half4 b4 = *reinterpret_cast<const half4*>(bl + a * 32 + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) init[4 * g + e] = static_cast<float>(b4[e]);
            }
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[SET][a][b] = init;
        }
    };

    // ---- the epilogue of a finished tile, cut into steps. Unit u = (mt, np) = (u>>1, u&1) produces
    //      8 output channels of 64 pixels from acc[2np][mt] and acc[2np+1][mt]; piece pi of a unit
    //      handles elements 8*(pi&1).. of acc[2np + ((pi>>1)&1)][mt].
    float pv[8], pf[8];
    float4 pc[8];
    float sums[2][4];
    auto piece_issue = [&](auto set_c, auto pi_c) {          // index math + table gathers of piece pi
        constexpr int SET = decltype(set_c)::value, PI = decltype(pi_c)::value;
        constexpr int U = PI / 4, MTI = U >> 1, NP = U & 1, H = (PI >> 1) & 1, HALF = PI & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = acc[SET][2 * NP + H][MTI][8 * HALF + e];
            float t = fmaf(v, 16.0f, 128.0f);
            t = fminf(fmaxf(t, 0.0f), 255.99998f);
            pv[e] = v;
            pf[e] = __builtin_amdgcn_fractf(t);
            pc[e] = tab[static_cast<int>(t) * R];
        }
    };
    auto piece_finish = [&](auto pi_c) {                     // polynomial + partial chunk sums of piece pi
        constexpr int PI = decltype(pi_c)::value;
        constexpr int H = (PI >> 1) & 1, HALF = PI & 1;
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float q = fmaf(pc[e].w, pf[e], pc[e].z);
            q = fmaf(q, pf[e], pc[e].y);
            q = fmaf(q, pf[e], pc[e].x);
            z[e] = pv[e] * q;
        }
        sums[H][2 * HALF] = ((z[0] + z[1]) + z[2]) + z[3];
        sums[H][2 * HALF + 1] = ((z[4] + z[5]) + z[6]) + z[7];
    };
    auto combine = [&](auto u_c) {                           // 8 channels x 64 pixels -> output tile in LDS
        constexpr int U = decltype(u_c)::value, MTI = U >> 1, NP = U & 1;
        half8 o;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sums[0][g]), __float_as_uint(sums[1][g]), false, false);
            o[2 * g] = to_half(__uint_as_float(sw[0]));
            o[2 * g + 1] = to_half(__uint_as_float(sw[1]));
        }
        const int row = (wm * MT + MTI) * 32 + (lane & 31);
        *reinterpret_cast<half8*>(oaddr(row, wn * 4 + NP * 2 + hi)) = o;
    };
    auto store_tile = [&](int n_i) {                         // output tile -> memory, whole 128-B lines
#pragma unroll
        for (int j = 0; j < BM * 8 / NTHREADS; ++j) {
            const int u = j * NTHREADS + tid;
            const int row = u >> 3, ch = u & 7;
            if (m0 + row < p.M) {
                store_line(p.y + static_cast<size_t>(m0 + row) * p.ldy + n_i * BNO + ch * 8,
                           *reinterpret_cast<const half8*>(oaddr(row, ch)));
            }
        }
    };
    // step q of the previous tile's epilogue (compile-time schedule)
    auto epilogue_step = [&](auto set_c, auto q_c, int n_i) {
        constexpr int Q = decltype(q_c)::value;
        constexpr int PI = piece_at<NK>(Q), CU = combine_at<NK>(Q);
        if constexpr (PI >= 0) {
            if constexpr (PI % 4 != 0) piece_finish(std::integral_constant<int, (PI > 0 ? PI - 1 : 0)>{});
            piece_issue(set_c, std::integral_constant<int, (PI >= 0 ? PI : 0)>{});
        }
        if constexpr (CU >= 0) {
            piece_finish(std::integral_constant<int, (CU >= 0 ? 4 * CU + 3 : 0)>{});
            combine(std::integral_constant<int, (CU >= 0 ? CU : 0)>{});
        }
        if constexpr (store_at<NK>(Q)) {
            if (n_i >= 2) store_tile(n_i - 2);
        }
    };

    // ---- main loop of tile n_i into accumulator set CUR, epilogue of tile n_i-1 (set CUR^1) inside.
    //      t (k-step) and s (k-slice) are compile-time constants: the epilogue schedule is static.
    auto tile_body = [&](auto cur_c, auto prev_c, int n_i) {
        constexpr int CUR = decltype(cur_c)::value;
        constexpr bool HAS_PREV = decltype(prev_c)::value;
        const int n_next = min(n_i + 1, tiles_n - 1);        // last tile: a harmless reload, no branch in the loop
        static_for<NK>([&](auto t_c) {
            constexpr int T = decltype(t_c)::value;
            __syncthreads();                                  // step landed (vmcnt(0)), the other buffer is free
            constexpr int cur = T & 1;                        // NK is even: parity of n_i*NK + t
            const char* xs = smem + cur * STAGE_BYTES + wm * (MT * 32 * 128);
            const char* ws = smem + cur * STAGE_BYTES + XT_BYTES + wn * (NT * 32 * 128);
            half8 xf[2][MT], wf[2][NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[0][i] = *reinterpret_cast<const half8*>(xs + i * (32 * 128) + foff[0]);
#pragma unroll
            for (int i = 0; i < NT; ++i) wf[0][i] = *reinterpret_cast<const half8*>(ws + i * (32 * 128) + foff[0]);
            static_for<4>([&](auto s_c) {
                constexpr int S = decltype(s_c)::value;
                if constexpr (S < 3) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        xf[(S + 1) & 1][i] = *reinterpret_cast<const half8*>(xs + i * (32 * 128) + foff[S + 1]);
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        wf[(S + 1) & 1][i] = *reinterpret_cast<const half8*>(ws + i * (32 * 128) + foff[S + 1]);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[CUR][nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[S & 1][nt], xf[S & 1][mt], acc[CUR][nt][mt], 0, 0, 0);
                if constexpr (T + 1 < NK) stage(cur ^ 1, n_i, (T + 1) * BK, S);
                else                      stage(cur ^ 1, n_next, 0, S);
                if constexpr (HAS_PREV) {
                    epilogue_step(std::integral_constant<int, CUR ^ 1>{}, std::integral_constant<int, T * 4 + S>{}, n_i);
                }
                if constexpr (SCHED) {
                    // one MFMA, then the VALU / LDS work that fits under its 32 cycles
#pragma unroll
                    for (int i = 0; i < NT * MT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // DS read
                        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);     // VALU
                    }
                }
            });
        });
        // set CUR^1 is consumed: it takes the next tile's bias
        init_acc(std::integral_constant<int, CUR ^ 1>{}, n_next);
    };

    __syncthreads();                                          // bias / table in LDS
    init_acc(std::integral_constant<int, 0>{}, 0);
    tile_body(std::integral_constant<int, 0>{}, std::false_type{}, 0);
    for (int n_i = 1; n_i < tiles_n; n_i += 2) {
        tile_body(std::integral_constant<int, 1>{}, std::true_type{}, n_i);
        if (n_i + 1 < tiles_n) tile_body(std::integral_constant<int, 0>{}, std::true_type{}, n_i + 1);
    }

    // ---- tail: the tile before the last one leaves, the last tile's epilogue runs on its own
    __syncthreads();
    if (tiles_n >= 2) store_tile(tiles_n - 2);
    __syncthreads();
    auto tail = [&](auto set_c) {
#define DCVC_PIPE_TAIL(U)                                                                                   \
        piece_issue(set_c, std::integral_constant<int, 4 * U>{});                                           \
        piece_finish(std::integral_constant<int, 4 * U>{});                                                 \
        piece_issue(set_c, std::integral_constant<int, 4 * U + 1>{});                                       \
        piece_finish(std::integral_constant<int, 4 * U + 1>{});                                             \
        piece_issue(set_c, std::integral_constant<int, 4 * U + 2>{});                                       \
        piece_finish(std::integral_constant<int, 4 * U + 2>{});                                             \
        piece_issue(set_c, std::integral_constant<int, 4 * U + 3>{});                                       \
        piece_finish(std::integral_constant<int, 4 * U + 3>{});                                             \
        combine(std::integral_constant<int, U>{});
        DCVC_PIPE_TAIL(0) DCVC_PIPE_TAIL(1) DCVC_PIPE_TAIL(2) DCVC_PIPE_TAIL(3)
#undef DCVC_PIPE_TAIL
    };
    if ((tiles_n - 1) & 1) tail(std::integral_constant<int, 1>{});
    else                   tail(std::integral_constant<int, 0>{});
    __syncthreads();
    store_tile(tiles_n - 1);
}

template <int NK>
void launch(const PipeParams& p, hipStream_t stream)
{
    static const bool sched = [] { const char* e = getenv("DCVC_GEMM_PIPE_SCHED"); return e == nullptr || atoi(e) != 0; }();
    const int smem_bytes = OFF_BIAS + p.N * 2;
    const dim3 grid((p.M + BM - 1) / BM), block(NTHREADS);
    auto go = [&](auto kern) {
        static std::once_flag once;
        std::call_once(once, [&] {
            hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          OFF_BIAS + MAX_N * 2), "hipFuncSetAttribute(gemm_pipe)");
        });
        hipLaunchKernelGGL(kern, grid, block, smem_bytes, stream, p);
    };
    if (sched) go(gemm_pipe_kernel<NK, true>);
    else       go(gemm_pipe_kernel<NK, false>);
    hip_check(hipGetLastError(), "gemm_pipe launch");
}

}  // namespace

bool gemm_pipe_supported(int pixels, int cin, int cout)
{
    return (cin == 384 || cin == 512) && cout % BN == 0 && cout <= MAX_N && pixels >= 128 * 192;
}

bool gemm_pipe_enabled()
{
    static const bool on = [] { const char* e = getenv("DCVC_GEMM_PIPE"); return e != nullptr && atoi(e) != 0; }();
    return on;
}

void conv1x1_wsilu_chunk_pipe(const Conv1x1Desc& d, hipStream_t stream)
{
    if (!d.wsilu || !d.chunk_add || !d.bias || d.r1 || d.r2 || d.q || d.q2) {
        throw std::invalid_argument("gemm_pipe: conv1x1 + bias + WSiLU + chunk-add only");
    }
    if (!((d.cin == 384 || d.cin == 512) && d.cout % BN == 0 && d.cout <= MAX_N) || d.pixels <= 0 || (d.ldx % 8) || (d.ldy % 8)) {
        throw std::invalid_argument("gemm_pipe: unsupported shape");
    }
    PipeParams p{};
    p.x = d.x; p.ldx = d.ldx; p.w = d.w; p.bias = d.bias; p.wsilu = wsilu_table_device();
    p.y = d.y; p.ldy = d.ldy; p.M = d.pixels; p.N = d.cout;
    if (d.cin == 384) launch<6>(p, stream);
    else              launch<8>(p, stream);
}

}  // namespace dcvc
