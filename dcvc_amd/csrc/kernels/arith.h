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
// Arithmetic policy v2: sigmoid(4 v) is a piecewise cubic on 256 segments of width 1/16 over
// [-8, 8) (max abs error 6.2e-7, three orders of magnitude below the fp16 resolution of the
// result), evaluated with exactly-rounded operations only (fmaf, floor, min/max) so that the CPU
// oracle reproduces it bit for bit. It replaces an exp + IEEE-division formulation that cost
// ~30 VALU operations per element - more than the matrix-core time of the 4x expanded FFN tile.
// `tab` points to the coefficient table (kernels/wsilu_table.h) in LDS, entry e at tab[e * R]:
// R = 1 is the plain 4 KiB table; the contraction kernel's epilogue uses R interleaved copies
// with every lane reading "its own" copy (tab already offset by lane & (R - 1)), which puts the 16
// lanes of a ds_read_b128 group on 16 different 16-byte bank slots whatever entries they ask for.
template <int R = 1>
__device__ __forceinline__ float wsilu_spec(float v, const float4* tab)
{
    float t = fmaf(v, 16.0f, 128.0f);
    t = fminf(fmaxf(t, 0.0f), 255.99998f);
    const float f = __builtin_amdgcn_fractf(t);      // t - floor(t), exact
    const float4 c = tab[static_cast<int>(t) * R];    // t >= 0: truncation == floor
    float p = fmaf(c.w, f, c.z);
    p = fmaf(p, f, c.y);
    p = fmaf(p, f, c.x);
    return v * p;
}

// Batched forms: all table indices first, then all LDS reads, then the polynomials - the loads
// overlap instead of exposing one LDS round trip per element.
template <int R = 1>
__device__ __forceinline__ void wsilu8(float (&v)[8], const float4* tab)
{
    float f[8];
    float4 c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = fmaf(v[e], 16.0f, 128.0f);
        t = fminf(fmaxf(t, 0.0f), 255.99998f);
        f[e] = __builtin_amdgcn_fractf(t);
        c[e] = tab[static_cast<int>(t) * R];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float p = fmaf(c[e].w, f[e], c[e].z);
        p = fmaf(p, f[e], c[e].y);
        p = fmaf(p, f[e], c[e].x);
        v[e] = v[e] * p;
    }
}

template <int R = 1>
__device__ __forceinline__ void wsilu16(const float16v& a, float (&z)[16], const float4* tab)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = a[8 * h + e];
        wsilu8<R>(v, tab);
#pragma unroll
        for (int e = 0; e < 8; ++e) z[8 * h + e] = v[e];
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
