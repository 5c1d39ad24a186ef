// glds_fill.hip - how fast can a CU fill LDS through global_load_lds when the loads are issued
// (a) in one burst per k-step and drained with vmcnt(0) (the contraction kernel's 2-stage loop) or
// (b) as a deep pipeline with counted vmcnt (DEPTH slabs in flight)?  No MFMA work: pure fill rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o glds_fill glds_fill.hip ; run: ./glds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ROWS rows per slab, RB bytes per row and slab (64 or 128), DEPTH slabs in flight, NT threads
template <int ROWS, int RB, int DEPTH, int NT>
__global__ void __launch_bounds__(NT) fill_kernel(const char* src, int row_stride, int k_slabs, int tiles_m,
                                                  long long* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLAB = ROWS * RB;
    constexpr int UNITS = SLAB / 16 / NT;              // 16-B units per thread and slab
    constexpr int CPR = RB / 16;                       // chunks per row
    const int tid = threadIdx.x, wave = tid >> 6;
    const int tile = blockIdx.x % tiles_m;
    const char* base = src + static_cast<size_t>(tile) * ROWS * row_stride;
    auto issue = [&](int slab) {
        char* dst = smem + (slab % DEPTH) * SLAB;
#pragma unroll
        for (int j = 0; j < UNITS; ++j) {
            const int u = j * NT + tid;
            const int row = u / CPR, ch = u % CPR;
            const char* p = base + static_cast<size_t>(row) * row_stride + slab * RB + ch * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(dst + (j * NT + wave * 64) * 16), 16, 0, 0);
        }
    };
    const long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < DEPTH - 1 && s < k_slabs; ++s) issue(s);
    for (int t = 0; t < k_slabs; ++t) {
        if (t + DEPTH - 1 < k_slabs) {
            if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * UNITS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + DEPTH - 1 < k_slabs) issue(t + DEPTH - 1);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int ROWS, int RB, int DEPTH>
void run(const char* name, const char* src, int row_stride, int rows_total, int k_bytes, long long* dout)
{
    constexpr int NT = 512;
    const int k_slabs = k_bytes / RB, tiles_m = rows_total / ROWS;
    const int blocks = 256;
    const int smem = DEPTH * ROWS * RB;
    auto kern = fill_kernel<ROWS, RB, DEPTH, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), smem, 0, src, row_stride, k_slabs, tiles_m, dout);
    }
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), dout, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += v;
    const double cyc = sum / blocks;
    const double bytes = static_cast<double>(k_slabs) * ROWS * RB;
    printf("%-44s slab %3d KB x %3d  in flight %3d KB : %8.0f cycles/WG  %6.1f B/clk/CU\n", name, ROWS * RB / 1024,
           k_slabs, (DEPTH - 1) * ROWS * RB / 1024, cyc, bytes / cyc);
}

int main()
{
    const int rows_total = 32768 + 8192, k_bytes = 6144;      // 40960 rows x 6 KB = 240 MB (> L2, < MALL)
    char* src;
    long long* dout;
    hipMalloc(&src, static_cast<size_t>(rows_total) * k_bytes);
    hipMemset(src, 1, static_cast<size_t>(rows_total) * k_bytes);
    hipMalloc(&dout, 4096 * sizeof(long long));
    // 512 rows per slab stands for X tile (256 rows) + W tile (256 rows) of the 256x256 contraction tile
    run<512, 128, 2>("burst, 128-B rows, 2 stages (current loop)", src, k_bytes, rows_total, k_bytes, dout);
    run<512, 64, 2>("burst, 64-B rows, 2 stages", src, k_bytes, rows_total, k_bytes, dout);
    run<512, 64, 3>("pipelined, 64-B rows, 3 stages", src, k_bytes, rows_total, k_bytes, dout);
    run<512, 64, 4>("pipelined, 64-B rows, 4 stages", src, k_bytes, rows_total, k_bytes, dout);
    run<256, 128, 4>("pipelined, 128-B rows, 4 stages, half slabs", src, k_bytes, rows_total, k_bytes, dout);
    run<512, 128, 2>("(again) burst, 128-B rows, 2 stages", src, k_bytes, rows_total, k_bytes, dout);
    // L2-resident source: every WG reads the same 512 rows (the weight-like operand)
    run<512, 128, 2>("burst, 128-B rows, L2-resident", src, k_bytes, 512, k_bytes, dout);
    run<512, 64, 4>("pipelined, 64-B rows, 4 stages, L2-resident", src, k_bytes, 512, k_bytes, dout);
    return 0;
}
