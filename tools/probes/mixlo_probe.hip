// mixlo_probe.hip - does v_fma_mixlo_f16 (what hipcc emits for (half)(a * b) with fp32 a, b) round once
// (exact product -> fp16) or twice (fp32, then fp16)? Compares it with v_mul_f32 + v_cvt_f16_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half_t;
__global__ void k(const float* a, const float* b, unsigned short* fused, unsigned short* split, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    half_t f, s;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(f) : "v"(a[i]), "v"(b[i]));
    float p;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(a[i]), "v"(b[i]));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(s) : "v"(p));
    fused[i] = *reinterpret_cast<unsigned short*>(&f);
    split[i] = *reinterpret_cast<unsigned short*>(&s);
}
int main()
{
    const int n = 1 << 24;
    std::vector<float> a(n), b(n);
    srand(1);
    for (int i = 0; i < n; ++i) { a[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f; b[i] = rand() / (float)RAND_MAX; }
    float *da, *db; unsigned short *df, *ds;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&df, n * 2); hipMalloc(&ds, n * 2);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, df, ds, n);
    std::vector<unsigned short> f(n), s(n);
    hipMemcpy(f.data(), df, n * 2, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 2, hipMemcpyDeviceToHost);
    long diff = 0, host_split = 0, host_fused = 0;
    for (int i = 0; i < n; ++i) {
        diff += f[i] != s[i];
        const float p32 = a[i] * b[i];
        const half_t hs = (half_t)p32;
        const half_t hf = (half_t)((double)a[i] * (double)b[i]);     // exact product (48 bits fit a double), one rounding
        host_split += *reinterpret_cast<const unsigned short*>(&hs) != s[i];
        host_fused += *reinterpret_cast<const unsigned short*>(&hf) != f[i];
    }
    printf("n = %d: fused != split on %ld elements; split != host double-rounding on %ld; fused != host single-rounding on %ld\n",
           n, diff, host_split, host_fused);
    return 0;
}
