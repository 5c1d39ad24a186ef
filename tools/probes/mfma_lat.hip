// Cycles per v_mfma_f32_32x32x16_f16 for a lone wave per SIMD: N independent accumulators rotated,
// accumulators in VGPRs ("v") or AGPRs ("a").
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lat tools/probes/mfma_lat.hip && ./mfma_lat
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int N, bool AGPR>
__global__ void __launch_bounds__(256, 1) chain(long long* out, float* sink)
{
    float16v acc[N];
    for (int i = 0; i < N; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)(e * 0.01f); }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                else      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0.f;
    for (int i = 0; i < N; ++i) s += acc[i][0] + acc[i][7];
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;
}

template <int N, bool AGPR>
void run(long long* d, float* sink)
{
    hipLaunchKernelGGL((chain<N, AGPR>), dim3(256), dim3(256), 0, 0, d, sink);
    hipLaunchKernelGGL((chain<N, AGPR>), dim3(256), dim3(256), 0, 0, d, sink);
    hipDeviceSynchronize();
    long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%2d independent accumulators in %s: %6.1f cycles per MFMA\n", N, AGPR ? "AGPRs" : "VGPRs", (double)h / (64.0 * 4 * N));
}

int main()
{
    long long* d; float* sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 64);
    run<1, false>(d, sink); run<2, false>(d, sink); run<4, false>(d, sink); run<8, false>(d, sink);
    run<1, true>(d, sink); run<2, true>(d, sink); run<4, true>(d, sink); run<8, true>(d, sink); run<12, true>(d, sink);
    return 0;
}
