// mfma_probe2 - generic trial runner for v_mfma_f32_32x32x16_f16 arithmetic experiments.
// Input file : int32 T, then T x { half a[16], half b[16], float c }
// Output file: T x { float d(0,0), int32 n_lanes_differing }
// Every row of A is a[], every column of B is b[], C is constant c -> all outputs equal.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

struct Trial {
    _Float16 a[16], b[16];
    float c;
};

__global__ void run(const Trial* __restrict__ tr, float* __restrict__ d, int* __restrict__ mism)
{
    const Trial& t = tr[blockIdx.x];
    const int kh = threadIdx.x >> 5;
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = t.a[kh * 8 + j];
        b[j] = t.b[kh * 8 + j];
    }
    float16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = t.c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    const float d0 = __shfl(acc[0], 0);
    int bad = 0;
    for (int r = 0; r < 16; ++r) bad += (__float_as_uint(acc[r]) != __float_as_uint(d0));
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if (threadIdx.x == 0) {
        d[blockIdx.x] = d0;
        mism[blockIdx.x] = bad;
    }
}

int main(int argc, char** argv)
{
    if (argc < 3) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t T = 0;
    if (fread(&T, 4, 1, f) != 1) return 2;
    std::vector<Trial> h(T);
    if (fread(h.data(), sizeof(Trial), T, f) != (size_t)T) return 2;
    fclose(f);
    Trial* dt; float* dd; int* dm;
    hipMalloc(&dt, sizeof(Trial) * T); hipMalloc(&dd, 4 * T); hipMalloc(&dm, 4 * T);
    hipMemcpy(dt, h.data(), sizeof(Trial) * T, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(run, dim3(T), dim3(64), 0, 0, dt, dd, dm);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 3; }
    std::vector<float> d(T); std::vector<int> m(T);
    hipMemcpy(d.data(), dd, 4 * T, hipMemcpyDeviceToHost);
    hipMemcpy(m.data(), dm, 4 * T, hipMemcpyDeviceToHost);
    f = fopen(argv[2], "wb");
    for (int i = 0; i < T; ++i) { fwrite(&d[i], 4, 1, f); fwrite(&m[i], 4, 1, f); }
    fclose(f);
    printf("mfma_probe2: %d trials\n", T);
    return 0;
}
