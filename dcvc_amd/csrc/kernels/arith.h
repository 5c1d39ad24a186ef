// Arithmetic policy of the MI355X codec (device side). DESIGN.md §"Arithmetic policy".
//
// Every tensor that the reference proxy materialises (one per op of SURVEY §2.3 / §2.4) is an
// fp16 NHWC tensor here too; inside an op the math is fp32 with every operation spelled out
// (fmaf / + / * / IEEE division, no fast-math, no contraction) so that the CPU oracle
// (oracle/nn_oracle.c) reproduces it bit for bit.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace dcvc {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

// WSiLU(v) = v * sigmoid(4 v)   (reference: layers.py:106-111; the CUDA epilogue computes it in
// fp32 with fast-math exp/div, conv1x1_kernel.h:32-35).
// Arithmetic policy v3 (round 3): sigma4(v) = sigmoid(4 v) is a piecewise QUADRATIC in v itself (global
// coordinate, no per-segment fraction) on 256 segments of width 1/32 over [-4, 4):
//     vc  = med3(v, -4, 4 - 2^-9)                      (only the segment choice is clamped)
//     x   = vc + 4100.0f                               (4100 = 2^12 + 4: one ulp of x is 2^-11)
//     seg = bits(x)[13:6]                              (= floor((vc + 4) * 32) of the ROUNDED sum)
//     p   = fmaf(fmaf(c2, v, c1), v, c0)               (c = table row seg; rows 0 and 255 are the
//                                                        constants 0 and 1, so v outside the range is safe)
//     WSiLU(v) = v * p            chunk-add of 4:  s = fmaf(v3, p3, fmaf(v2, p2, fmaf(v1, p1, v0 * p0)))
// 6 VALU operations per element (v_med3, v_add, v_and_or, 2 v_fma, v_mul / v_fma) and one 16-byte LDS
// gather, against ~10 for policy v2's cubic in the segment fraction (fma, max, min, fract, cvt, shift-add,
// 3 fma, mul, add): the epilogue of the 4x expanded FFN tile costs as much VALU issue time as the tile's
// MFMAs, so this is a first-order term of the block kernels. Max abs error of p 1.5e-6, of
// WSiLU 5.1e-7 (tools/gen_wsilu_table.py; policy v2: 6.2e-7 / 2.5e-7) - far below the fp16 resolution of
// the stored result.
// Only exactly rounded operations: the CPU oracle (oracle/nn_oracle.c) reproduces it bit for bit.
// `tab` points to the coefficient table (kernels/wsilu_table.h, rows {c0, c1, c2, 0}) in LDS, entry e at
// tab[e * R]: R = 1 is the plain 4 KiB table; the contraction kernels' epilogues use R interleaved copies
// with every lane reading "its own" copy (tab already offset by lane & (R - 1)), which spreads the 16
// lanes of a ds_read_b128 group over 16-byte bank slots whatever entries they ask for.
constexpr int kArithPolicyVersion = 3;       // dcvc_arith_policy_version(): bump with ANY change that moves a stored fp16 / symbol
constexpr float kWsiluLo = -4.0f;
constexpr float kWsiluHi = 3.998046875f;     // 4 - 2^-9
constexpr float kWsiluMagic = 4100.0f;       // 2^12 + 4

// byte offset of the table row of `v` inside a table of R interleaved copies (row e, copy 0 at e * 16 R)
template <int R>
__device__ __forceinline__ unsigned wsilu_row_offset(float v)
{
    const float vc = __builtin_amdgcn_fmed3f(v, kWsiluLo, kWsiluHi);
    const unsigned b = __float_as_uint(vc + kWsiluMagic) & 0x3fc0u;      // seg * 64
    if constexpr (R >= 4) return b * (R / 4);
    else return b / (4 / R);
}

template <int R>
__device__ __forceinline__ float4 wsilu_row(float v, const float4* tab)
{
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tab) + wsilu_row_offset<R>(v));
}

// The same from an LDS byte address (`lane_tab` = this lane's copy of row 0). DISJOINT: the caller
// guarantees that lane_tab has no bit in common with a row offset (R = 4: bits 6..13), so the sum is an OR
// and index mask + base fold into ONE v_and_or_b32.
template <int R, bool DISJOINT>
__device__ __forceinline__ float4 wsilu_row_lds(float v, unsigned lane_tab)
{
    const unsigned off = wsilu_row_offset<R>(v);
    const unsigned a = DISJOINT ? (off | lane_tab) : (off + lane_tab);
    const float4v r = *reinterpret_cast<const __attribute__((address_space(3))) float4v*>(a);
    return make_float4(r[0], r[1], r[2], r[3]);
}

__device__ __forceinline__ float wsilu_poly(float v, const float4 c)
{
    const float p = fmaf(fmaf(c.z, v, c.y), v, c.x);
    // Only three of the row's four floats are used, and hipcc then narrows the gather to ds_read_b96: 8 LDS
    // cycles per wave-instruction instead of ds_read_b128's 4 (MI355X_MICROARCH.md, LDS table). Naming the
    // fourth HERE, where the row has to have arrived anyway, keeps the load whole without an extra wait (the
    // same statement right behind the load made the wave sit out one LDS round trip per gather: measured).
    asm volatile("" ::"v"(c.w));
    return p;
}

template <int R = 1>
__device__ __forceinline__ float wsilu_spec(float v, const float4* tab)
{
    return v * wsilu_poly(v, wsilu_row<R>(v, tab));
}

// Batched forms: all table rows first (the LDS reads overlap instead of exposing one round trip per
// element), then the polynomials.
template <int R = 1>
__device__ __forceinline__ void wsilu8(float (&v)[8], const float4* tab)
{
    float4 c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = wsilu_row<R>(v[e], tab);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * wsilu_poly(v[e], c[e]);
}

// WSiLU + chunk-add of one accumulator tile: s[g] = sum over the 4 adjacent channels 4g .. 4g+3
// (conv1x1_bias_wsilu_chunk_add.cu:356-390), as one fma chain per group
template <int R = 1>
__device__ __forceinline__ void wsilu_chunk16(const float16v& a, float (&s)[4], const float4* tab)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 c[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) c[e] = wsilu_row<R>(a[8 * h + e], tab);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float acc = a[8 * h + 4 * g] * wsilu_poly(a[8 * h + 4 * g], c[4 * g]);
#pragma unroll
            for (int e = 1; e < 4; ++e) acc = fmaf(a[8 * h + 4 * g + e], wsilu_poly(a[8 * h + 4 * g + e], c[4 * g + e]), acc);
            s[2 * h + g] = acc;
        }
    }
}

// 16-byte store of a finished output line. (A write-through variant - global_store ... sc0 sc1 -
// was measured: +2.4 % on the intra bench because the end-of-kernel L2 write-back shrinks, but
// WRONG at 1080p: a later launch's plain loads can hit a stale line of a reused scratch buffer.
// Plain stores it is; the pairing rule is sc1 stores AND sc1 loads, MI355X_MICROARCH.md.)
__device__ __forceinline__ void store_line(half_t* p, half8 v)
{
    *reinterpret_cast<half8*>(p) = v;
}

// C round(): half away from zero (the reference's symbol kernels call round() on a float,
// elementwise/stream.cu:587-588,873-874).
__device__ __forceinline__ float round_half_away(float v)
{
    const float a = fabsf(v);
    const float f = floorf(a);
    const float r = (a - f >= 0.5f) ? f + 1.0f : f;
    return copysignf(r, v);
}

__device__ __forceinline__ half_t to_half(float v)
{
    // The empty asm makes `v` opaque: without it hipcc contracts (half)(a * b) into ONE
    // v_fma_mixlo_f16, which rounds the exact product straight to fp16 - a single rounding where the
    // arithmetic policy (and the oracle) have two, fp32 then fp16. Measured on MI355X
    // (tools/probes/mixlo_probe.hip): 973 of 16.7 M random products differ by one fp16 ulp.
    asm("" : "+v"(v));
    return static_cast<half_t>(v);   // v_cvt_f16_f32, round-to-nearest-even, subnormals kept
}

// result of a single fp16 operation a (op) b, evaluated as the correctly rounded fp32 result
// rounded again to fp16 - identical to a native fp16 operation (double rounding is innocuous for
// + - * when the wide format has >= 2*11+2 significand bits).
__device__ __forceinline__ half_t hadd(half_t a, half_t b)
{
    return to_half(static_cast<float>(a) + static_cast<float>(b));
}
__device__ __forceinline__ half_t hsub(half_t a, half_t b)
{
    return to_half(static_cast<float>(a) - static_cast<float>(b));
}
__device__ __forceinline__ half_t hmul(half_t a, half_t b)
{
    return to_half(static_cast<float>(a) * static_cast<float>(b));
}

}  // namespace dcvc
