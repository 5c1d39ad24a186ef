// Is a lone wave per SIMD instruction-fetch bound on straight-line code that is executed once?
// N MFMAs (8 B each) either fully unrolled (cold code, N*8 bytes) or as a rolled loop of 32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

// FILL: 6 independent 8-byte VALU instructions behind every MFMA (56 B per MFMA in total)
#define FILL asm volatile("v_fma_f32 %0, %0, 1.0, 0x3f800000\n\tv_fma_f32 %1, %1, 1.0, 0x3f800000\n\tv_fma_f32 %2, %2, 1.0, 0x3f800000\n\tv_fma_f32 %3, %3, 1.0, 0x3f800000\n\tv_fma_f32 %4, %4, 1.0, 0x3f800000\n\tv_fma_f32 %5, %5, 1.0, 0x3f800000" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5));
template <int N, bool UNROLLED>
__global__ void __launch_bounds__(256, 1) k(long long* out, float* sink)
{
    float16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float f0 = 1, f1 = 2, f2 = 3, f3 = 4, f4 = 5, f5 = 6;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)(e * 0.01f); }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (UNROLLED) {
#pragma unroll
        for (int i = 0; i < N; ++i) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b)); FILL }
    } else {
        for (int it = 0; it < N / 32; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i & 3]) : "v"(a), "v"(b)); FILL }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    s += f0 + f1 + f2 + f3 + f4 + f5;
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;
}

template <int N, bool U>
void run(long long* d, float* sink)
{
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<N, U>), dim3(256), dim3(256), 0, 0, d, sink);
    hipDeviceSynchronize();
    long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%5d MFMAs %s (%3d KB of code): %6.1f cycles per MFMA\n", N, U ? "straight-line" : "loop of 32    ", U ? N * 80 / 1024 : 0, (double)h / N);
}

int main()
{
    long long* d; float* sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 64);
    run<1024, false>(d, sink); run<256, true>(d, sink); run<512, true>(d, sink); run<1024, true>(d, sink); run<2048, true>(d, sink);
    return 0;
}
