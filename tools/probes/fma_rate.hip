// fma_rate - issue rate of the fp32-accumulating multiply-adds the depthwise conv inside the block kernel can be built from
// (round 6): v_fma_mix_f32 (fp16 sources), v_fma_f32, v_pk_fma_f32, v_cvt_f32_f16. One wave per SIMD and two, cycles per instruction.
// build: hipcc -O2 --offload-arch=gfx950 tools/probes/fma_rate.hip -o tools/_bin/fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
#define OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); return 1; } } while (0)
constexpr int N = 4096, U = 16;

template <int MODE>
__global__ void __launch_bounds__(512) probe(float* out, long long* cyc, unsigned seed)
{
    float acc[U];
    float2v acc2[U / 2];
    for (int i = 0; i < U; ++i) acc[i] = static_cast<float>(threadIdx.x + i);
    for (int i = 0; i < U / 2; ++i) acc2[i] = float2v{acc[2 * i], acc[2 * i + 1]};
    unsigned a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
    float fa = __uint_as_float(0x3f800000u | (a & 0xffff)), fb = __uint_as_float(0x3f000000u | (b & 0xffff));
    float2v fa2 = {fa, fb}, fb2 = {fb, fa};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int n = 0; n < N / U; ++n) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (MODE == 0) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i / 2]) : "v"(fa2), "v"(fb2));      // (U/2 distinct accumulators, each twice)
            if (MODE == 3) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(acc[i]) : "v"(a));
            if (MODE == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(acc[i]) : "v"(a));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < U; ++i) s += acc[i];
    for (int i = 0; i < U / 2; ++i) s += acc2[i][0] + acc2[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char* name)
{
    float* out; long long* cyc;
    OK(hipMalloc(&out, 512 * 4)); OK(hipMalloc(&cyc, 8));
    for (int threads : {256, 512}) {      // one / two waves per SIMD
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, 12345u);
        OK(hipDeviceSynchronize());
        long long c = 0;
        OK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        printf("%-22s %d wave(s) per SIMD: %6.2f cycles per instruction and wave (%d instructions per wave)\n", name, threads / 256, static_cast<double>(c) / N, N);
    }
    return 0;
}

int main()
{
    return run<0>("v_fma_mix_f32") || run<1>("v_fma_f32") || run<2>("v_pk_fma_f32") || run<3>("v_cvt_f32_f16") || run<4>("v_cvt_f32_f16_sdwa");
}
