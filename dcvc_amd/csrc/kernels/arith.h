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

// exp(t) for t in [-80, 80], built only from exactly-rounded IEEE operations:
// Cody-Waite reduction by ln2 (two fmaf), degree-6 Horner polynomial (Cephes expf coefficients),
// scaling by 2^n through the exponent field.
__device__ __forceinline__ float exp_spec(float t)
{
    t = fminf(fmaxf(t, -80.0f), 80.0f);
    const float n = rintf(t * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, t);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    p = fmaf(p, r2, r);
    p = p + 1.0f;
    const int ni = static_cast<int>(n);
    return p * __int_as_float((ni + 127) << 23);
}

// WSiLU(v) = v * sigmoid(4 v) = v / (1 + exp(-4 v))   (reference: layers.py:106-111; the CUDA
// epilogue computes it in fp32 too, conv1x1_kernel.h:32-35)
__device__ __forceinline__ float wsilu_spec(float v)
{
    const float e = exp_spec(-4.0f * v);
    return v / (1.0f + e);
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
