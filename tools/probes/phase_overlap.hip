// Do the MFMA phase of one wave and the VALU phase of ANOTHER wave on the same SIMD overlap?
// (round 4: the premise of an 8-wave block kernel - two waves per SIMD, each alternating a contraction with its own
// epilogue; dcb_nsplit's lone wave per SIMD hides 13 - 40 % of an epilogue under its own MFMAs.)
//
// Every wave loops IT times over [M x v_mfma_f32_32x32x16_f16 on 4 rotating accumulators] [V x v_fma_f32 in 8 independent
// chains (+ G LDS gathers of 16 bytes at lane-dependent rows, waited for)]. One workgroup per CU:
//   4 waves (1 per SIMD) with (2 M, 2 V, 2 G) per iteration   vs   8 waves (2 per SIMD) with (M, V, G):
// the same work per SIMD. Prints cycles per iteration; overlap shows as  t(8 waves) -> max(MFMA, VALU) instead of the sum.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/phase_overlap.hip -o tools/_bin/phase_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int WAVES, int M, int V, int G, int PRIO>
__global__ void __launch_bounds__(64 * WAVES, 1) phases(long long* out, float* sink, int iters)
{
    __shared__ float4v table[1024];
    for (int i = threadIdx.x; i < 1024; i += 64 * WAVES) table[i] = float4v{0.5f, 0.25f, 0.125f, 0.f};
    __syncthreads();
    float16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)(e * 0.01f); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
    const float ca = 0.999f, cb = 0.001f;
    unsigned row = (threadIdx.x * 37u) & 1023u;
    const int wave = threadIdx.x >> 6;
    if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < M; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float4v r = table[row];
            asm volatile("" : "+v"(r));
            x[g & 7] += r[0];
            row = (row * 5u + 1u) & 1023u;
        }
#pragma unroll
        for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(ca), "v"(cb));
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][9];
    for (int i = 0; i < 8; ++i) s += x[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;      // the LAST wave of a workgroup counts (the host takes the max)
    if (s == 12345.678f) sink[0] = s;
}

template <int WAVES, int M, int V, int G, int PRIO>
double run(long long* d, float* sink)
{
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((phases<WAVES, M, V, G, PRIO>), dim3(256), dim3(64 * WAVES), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    static long long h[256 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) {
        long long m = 0;
        for (int w = 0; w < WAVES; ++w) m = h[i * 8 + w] > m ? h[i * 8 + w] : m;
        s += (double)m;
    }
    return s / 256 / iters;
}

template <int M, int V, int G>
void compare(long long* d, float* sink)
{
    const double t4 = run<4, 2 * M, 2 * V, 2 * G, 0>(d, sink);
    const double t8 = run<8, M, V, G, 0>(d, sink);
    const double t8p = run<8, M, V, G, 1>(d, sink);
    const double t8q = run<8, M, V, G, 2>(d, sink);
    const double t4m = run<4, 2 * M, 0, 0, 0>(d, sink);
    const double t4v = run<4, 1, 2 * V, 2 * G, 0>(d, sink);
    printf("per SIMD and iteration: %3d MFMAs (floor %5d cycles), %4d VALU, %3d gathers | 4 waves %7.0f | 8 waves %7.0f | 8 waves, second half at prio 1 %7.0f | "
           "8 waves, prio 1 inside the MFMA phase %7.0f | MFMAs alone %7.0f | VALU alone (1 wave / SIMD) %7.0f\n",
           2 * M, 2 * M * 32, 2 * V, 2 * G, t4, t8, t8p, t8q, t4m, t4v);
}

int main()
{
    long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 64);
    compare<48, 300, 0>(d, sink);
    compare<48, 300, 32>(d, sink);
    compare<48, 150, 0>(d, sink);
    compare<96, 600, 64>(d, sink);
    compare<24, 150, 16>(d, sink);
    compare<36, 450, 48>(d, sink);
    return 0;
}
