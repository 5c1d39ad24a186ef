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

#include <cstdlib>
#include <mutex>
#include <stdexcept>

namespace dcvc {

const float4* wsilu_table_device();      // conv_gemm.hip

namespace {

constexpr int C = 384;                   // block width
constexpr int NTHREADS = 256;
constexpr int BM = 128;                  // pixels per workgroup
constexpr int SLAB = 128 * 64 * 2;       // [128 channels][64 k] fp16 = 16 KB
constexpr int NS = 5;                    // ring slots (15 slabs per ffn super-chunk: slot indices stay static)
constexpr int LPS = SLAB / (NTHREADS * 16);   // global_load_lds per thread and slab = 4
constexpr int R = 4;                     // interleaved copies of the WSiLU table
constexpr int TABLE_BYTES = WSILU_SEGMENTS * 16;
constexpr int OFF_TABLE = NS * SLAB;
constexpr int OFF_STAGE = OFF_TABLE + R * TABLE_BYTES;
constexpr int STAGE_BYTES = 32 * 256;    // per wave: 32 pixels x 128 channels fp16
constexpr int SMEM_BYTES = OFF_STAGE + 4 * STAGE_BYTES;
constexpr int G_A = 18, G_B = 90, G_D = 18;      // slabs of dc.3 / ffn.0+ffn.2 / next dc.0
static_assert(15 % NS == 0, "a super-chunk of 15 slabs must map onto the same ring slots every time");

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
};

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
    const int m = m0 + px;                       // this lane's pixel
    const int mc = min(m, p.M - 1);              // clamped for loads
    const int G = G_A + G_B + (p.w1n != nullptr ? G_D : 0);

    // ---- weight stream: slab g -> (matrix, first row, first k). Every matrix has row stride C.
    auto slab_base = [&](int g) -> const half_t* {
        if (g < G_A) {      // dc.3: k-step outer, channel chunk inner
            return p.w3 + static_cast<size_t>(g % 3) * (128 * C) + (g / 3) * 64;
        }
        g -= G_A;
        if (g < G_B) {
            const int sc = g / 15, r = g - sc * 15;
            if (r < 12) {
                return p.w0 + static_cast<size_t>(sc * 256 + (r / 6) * 128) * C + (r % 6) * 64;
            }
            return p.w2 + static_cast<size_t>(r - 12) * (128 * C) + sc * 64;
        }
        g -= G_B;
        return p.w1n + static_cast<size_t>(g / 6) * (128 * C) + (g % 6) * 64;
    };
    // staging plan (conv_gemm.hip): 16-B unit u = j*256 + tid -> slab row u>>3, physical chunk u&7,
    // logical chunk = physical ^ ((row>>1)&7); the LDS image is lane-linear
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
    const int toff = srow * C + schunk * 8;      // + j * 32 rows
    auto issue_slab = [&](int g, int slot) {
        if (g < G) {
            const half_t* base = slab_base(g) + toff;
            char* dst = smem + slot * SLAB;
#pragma unroll
            for (int j = 0; j < LPS; ++j) {
                __builtin_amdgcn_global_load_lds((gptr_t)(base + j * (32 * C)),
                                                 (lptr_t)(dst + (j * NTHREADS + wave * 64) * 16), 16, 0, 0);
            }
        }
    };

    // ---- B fragments of this lane's pixel: k-slice i = channels 16 i + 8 hi .. + 7
    half8 bf[24];
    {
        const half_t* row = p.t2 + static_cast<size_t>(mc) * p.ldt + 8 * hi;
#pragma unroll
        for (int i = 0; i < 24; ++i) bf[i] = *reinterpret_cast<const half8*>(row + 16 * i);
    }
#pragma unroll
    for (int g = 0; g < NS - 1; ++g) issue_slab(g, g);

    // WSiLU table -> LDS, R interleaved copies (lane l gathers from copy l & (R-1))
    {
        float4* t = reinterpret_cast<float4*>(smem + OFF_TABLE);
        for (int i = tid; i < R * WSILU_SEGMENTS; i += NTHREADS) t[i] = p.wsilu[i / R];
    }
    const float4* tab = reinterpret_cast<const float4*>(smem + OFF_TABLE) + (lane & (R - 1));

    // A-fragment offsets inside a slab for the four 16-wide k slices (conv_gemm.hip)
    const int fsw = (px >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = px * 128 + (((s * 2 + hi) ^ fsw) << 4);

    // One slab: wait until it has landed, make sure every wave is done with the slot the next
    // prefetch overwrites, prefetch, then 4 k-slices x 4 channel tiles of MFMAs. b(s) = the B
    // fragment of k-slice s, acc(nt) = the accumulator of channel tile nt.
    // `slot` = g % NS, passed separately so that it stays a compile-time constant wherever the caller
    // knows it (g itself is a run-time value inside the rolled loops).
    auto slab_step = [&](int g, int slot, auto&& bfrag, float16v (&acc)[4]) {
        if (g + NS - 1 < G) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue_slab(g + NS - 1, (slot + NS - 1) % NS);
        const char* ws = smem + slot * SLAB;
        half8 wf[2][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[0][nt] = *reinterpret_cast<const half8*>(ws + nt * 4096 + foff[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    wf[(s + 1) & 1][nt] = *reinterpret_cast<const half8*>(ws + nt * 4096 + foff[s + 1]);
            }
            const half8 b = bfrag(s);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s & 1][nt], b, acc[nt], 0, 0, 0);
        }
    };

    // accumulators of 4 channel tiles (128 channels from `first`) initialised with the bias:
    // acc[nt][r] = channel first + 32 nt + 8 (r>>2) + 4 hi + (r&3)
    auto bias_init = [&](float16v (&acc)[4], const half_t* bias, int first) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const half_t* bp = bias + first + 32 * nt + 4 * hi;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const half4 b4 = *reinterpret_cast<const half4*>(bp + 8 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nt][4 * g4 + e] = static_cast<float>(b4[e]);
            }
        }
    };
    // accumulator tile -> the two 8-channel runs this lane owns after pairing the half-waves:
    // run pr = channels 32 nt + 16 pr + 8 hi .. + 7 of the lane's pixel (= B-fragment layout)
    auto runs_of = [&](const float16v& a, int pr, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * pr + e]),
                                                             __float_as_uint(a[8 * pr + 4 + e]), false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
        }
    };
    // 128 output channels of the wave's 32 pixels -> memory in whole 256-B runs, via the wave's
    // own staging area (no other wave touches it: wave-local ordering is enough)
    char* stg = smem + OFF_STAGE + wave * STAGE_BYTES;
    auto flush128 = [&](const half8 (&o)[8], half_t* dst, int ld, int first) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cidx = 2 * i + hi;                   // (4 nt + 2 pr + hi), i = 2 nt + pr
            *reinterpret_cast<half8*>(stg + px * 256 + ((cidx ^ (px & 15)) << 4)) = o[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int u = it * 64 + lane;
            const int row = u >> 4, c16 = u & 15;
            const half8 v = *reinterpret_cast<const half8*>(stg + row * 256 + ((c16 ^ (row & 15)) << 4));
            if (m0 + row < p.M) {
                store_line(dst + static_cast<size_t>(m0 + row) * ld + first + c16 * 8, v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    int g = 0;
    // ================================================================ dc.3: y1 = W3 t2 + b3' + x
    // All 12 channel tiles accumulate at once (k-step outer, channel chunk inner) in the registers
    // ffn.2's accumulators take over afterwards; the fp16 result then replaces t2 in `bf` in place.
    float16v acc2[12];
    {
        float16v tmp[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            bias_init(tmp, p.b3, 128 * c);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc2[4 * c + nt] = tmp[nt];
        }
    }
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float16v a4[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) a4[nt] = acc2[4 * c + nt];
            slab_step(g, (3 * ks + c) % NS, [&](int s) { return bf[ks * 4 + s]; }, a4);
            ++g;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc2[4 * c + nt] = a4[nt];
        }
    }
    {
        const half_t* xrow = p.x + static_cast<size_t>(mc) * p.ldx + 8 * hi;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            half8 xr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[i] = *reinterpret_cast<const half8*>(xrow + 128 * c + 16 * i);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float v[8];
                    runs_of(acc2[4 * c + nt], pr, v);
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = to_half(v[e] + static_cast<float>(xr[2 * nt + pr][e]));
                    // opaque: otherwise hipcc keeps every element a second time, unpacked, for the
                    // residual add of the ffn.2 epilogue (192 values spilled across the ffn walk)
                    asm volatile("" : "+v"(o));
                    bf[8 * c + 2 * nt + pr] = o;
                }
        }
    }

    // ================================================================ ffn.0 -> t -> ffn.2 accumulators
    {
        float16v tmp[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            bias_init(tmp, p.b2, 128 * c);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc2[4 * c + nt] = tmp[nt];
        }
    }
    for (int sc = 0; sc < 6; ++sc) {
        half8 t3[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float16v acc[4];
            bias_init(acc, p.b0, sc * 256 + h * 128);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                slab_step(g, (G_A + 6 * h + ks) % NS, [&](int s) { return bf[ks * 4 + s]; }, acc);   // 15 % NS == 0
                ++g;
            }
            // z = wsilu(acc); sum of 4 adjacent channels ((z0+z1)+z2)+z3; pair the half-waves
#pragma unroll
            for (int np = 0; np < 2; ++np) {
                float sum[2][4];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float z[16];
                    wsilu16<R>(acc[2 * np + hh], z, tab);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        sum[hh][g4] = ((z[4 * g4] + z[4 * g4 + 1]) + z[4 * g4 + 2]) + z[4 * g4 + 3];
                }
                half8 o;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum[0][g4]),
                                                                     __float_as_uint(sum[1][g4]), false, false);
                    o[2 * g4] = to_half(__uint_as_float(sw[0]));
                    o[2 * g4 + 1] = to_half(__uint_as_float(sw[1]));
                }
                t3[2 * h + np] = o;
            }
        }
#pragma unroll
        for (int nc = 0; nc < 3; ++nc) {
            float16v a4[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) a4[nt] = acc2[4 * nc + nt];
            slab_step(g, (G_A + 12 + nc) % NS, [&](int s) { return t3[s]; }, a4);
            ++g;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc2[4 * nc + nt] = a4[nt];
        }
    }

    // ================================================================ ffn.2 epilogue: y
    {
        const half_t* xrow = p.x + static_cast<size_t>(mc) * p.ldx + 8 * hi;
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
                    if (p.shortcut) {
                        const half8 r8 = *reinterpret_cast<const half8*>(xrow + 16 * i);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(r8[e]);
                    }
                    if (p.q != nullptr) {
                        const half8 q8 = *reinterpret_cast<const half8*>(p.q + 16 * i + 8 * hi);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] * static_cast<float>(q8[e]);
                    }
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
                    if (p.q2 != nullptr) {
                        const half8 q8 = *reinterpret_cast<const half8*>(p.q2 + 16 * i + 8 * hi);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = hmul(o[e], q8[e]);
                    }
                    bf[i] = o;
                    o8[2 * nt + pr] = o;
                }
            flush128(o8, p.y, p.ldy, 128 * c);
        }
    }

    // ================================================================ dc.0 of the next block: t1' = WSiLU(W1' y + b1')
    if (p.w1n != nullptr) {
        for (int c = 0; c < 3; ++c) {
            float16v acc[4];
            bias_init(acc, p.b1n, 128 * c);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                slab_step(g, (G_A + G_B + 6 * c + ks) % NS, [&](int s) { return bf[ks * 4 + s]; }, acc);
                ++g;
            }
            half8 o8[8];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float v[8];
                    runs_of(acc[nt], pr, v);
                    wsilu8<R>(v, tab);
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
                    o8[2 * nt + pr] = o;
                }
            flush128(o8, p.t1n, p.ldt1, 128 * c);
        }
    }
}

}  // namespace

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
    static std::once_flag once;
    std::call_once(once, [] {
        hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(dcb_core_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES),
                  "hipFuncSetAttribute(dcb_core)");
    });
    hipLaunchKernelGGL(dcb_core_kernel, dim3((d.pixels + BM - 1) / BM), dim3(NTHREADS), SMEM_BYTES, stream, p);
    hip_check(hipGetLastError(), "dcb_core launch");
}

}  // namespace dcvc
