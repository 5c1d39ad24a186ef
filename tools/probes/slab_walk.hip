// slab_walk.hip - how fast can ONE CU walk a weight stream the way dcb_core does (16 KB slabs through a 5-slot LDS
// ring by LDS-DMA, one barrier per slab, 16 MFMAs per SIMD and slab on fragments read from the ring, activations
// resident in registers), with ONE wave per SIMD (4 waves, each all 4 tiles of a slab) or TWO (8 waves, the tiles of
// a slab split between the two waves of a SIMD), and with 0 / 4 / 8 independent VALU operations per MFMA standing in
// for the epilogue work. Prints shader-clock cycles per slab (MFMA floor: 16 x 32 = 512).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/slab_walk tools/probes/slab_walk.hip && tools/_bin/slab_walk
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int C = 384;
constexpr int SLAB = 16384;

__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

template <int WAVES, int VALU, int NS = 5, int MODE = 0>      // MODE 1: no MFMAs / fragment reads; 2: no LDS-DMA in the loop
__global__ void __launch_bounds__(WAVES * 64) walk(const half_t* __restrict__ w, int nslabs, float* __restrict__ out,
                                                   long long* __restrict__ cycles, int stagger)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = WAVES * 64;
    constexpr int LPS = SLAB / (NT * 16);                 // LDS-DMA pieces per thread and slab: 4 or 2
    constexpr int TILES = 16 / WAVES;                     // 32-row tiles of a slab per wave: 4 or 2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, hi = lane >> 5;
    const int tile0 = WAVES == 4 ? 0 : 2 * (wave >> 2);   // 8 waves: waves 0..3 take tiles 0,1; waves 4..7 tiles 2,3
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) void*)smem));

    // source-side swizzle of a wide slab [128 rows][64 k]: 16-B unit u = j * NT + tid -> row u >> 3, chunk (u & 7) ^ ((row >> 1) & 7)
    // stagger: workgroup b starts its walk `stagger * b` slabs into the (cyclic) stream, so that the 32 CUs of an XCD do
    // not ask their L2 for the same lines at the same time
    const int g0 = stagger * static_cast<int>(blockIdx.x);
    auto issue = [&](int gg, int j) {
        const int g = gg + g0;
        const int u = j * NT + tid;
        const int row = u >> 3, chunk = (u & 7) ^ ((row >> 1) & 7);
        const half_t* src = w + static_cast<size_t>(g % 24) * (128 * C) % (4 * C * C) + row * C + ((g / 24) % 6) * 64 + chunk * 8;
        lds_dma16(src, lds_base + (gg % NS) * SLAB + j * (NT * 16) + wave * 1024);
    };
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = (px * 128 + (((s * 2 + hi) ^ ((px >> 1) & 7)) << 4)) & 0xfff;

    half8 b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) b[s][e] = static_cast<half_t>(0.001f * ((lane * 7 + s * 3 + e) % 13));
    float16v acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + lane * 0.001f + i;

#pragma unroll
    for (int g = 0; g < NS - 1; ++g)
#pragma unroll
        for (int j = 0; j < LPS; ++j) issue(g, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
    __builtin_amdgcn_s_barrier();
    half8 head[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) head[t] = *reinterpret_cast<const half8*>(smem + foff[0] + (tile0 + t) * 4096);
    const long long t0 = __builtin_readcyclecounter();

    for (int g = 0; g < nslabs; ++g) {
        if (MODE != 2 && MODE != 4 && MODE != 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * LPS) : "memory");       // slab g+1 landed, NS-3 younger ones may fly
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE != 4) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* ws = smem + (g % NS) * SLAB;
        half8 wf[2][TILES];
        if (MODE == 3) {
            // MODE 3: the first fragments of slab g were read at the end of step g-1 (slab g was certified by THAT
            // step's barrier), into head[]
#pragma unroll
            for (int t = 0; t < TILES; ++t) wf[0][t] = head[t];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) {
#pragma unroll
                    for (int t = 0; t < TILES; ++t) wf[(s + 1) & 1][t] = *reinterpret_cast<const half8*>(ws + foff[s + 1] + (tile0 + t) * 4096);
                } else {
                    const char* wn = smem + ((g + 1) % NS) * SLAB;
#pragma unroll
                    for (int t = 0; t < TILES; ++t) head[t] = *reinterpret_cast<const half8*>(wn + foff[0] + (tile0 + t) * 4096);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s & 1][t], b[s], acc[t], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < VALU; ++i) v[i % 8] = fmaf(v[i % 8], 1.0001f, 0.25f);
                }
                if (LPS == 4 || (s & 1)) issue(g + NS - 1, LPS == 4 ? s : s >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
        if (MODE == 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (LPS == 4 || (s & 1)) issue(g + NS - 1, LPS == 4 ? s : s >> 1);
            }
            continue;
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) wf[0][t] = *reinterpret_cast<const half8*>(ws + foff[0] + (tile0 + t) * 4096);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3 && MODE != 5) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) wf[(s + 1) & 1][t] = *reinterpret_cast<const half8*>(ws + foff[s + 1] + (tile0 + t) * 4096);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(MODE == 5 ? wf[0][t] : wf[s & 1][t], b[s], acc[t], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < VALU; ++i) v[i % 8] = fmaf(v[i % 8], 1.0001f, 0.25f);
            }
            // the slab four steps ahead, a quarter (4 waves: one piece; 8 waves: a piece every other slice) at a time
            if (MODE != 2 && MODE != 4 && MODE != 5 && (LPS == 4 || (s & 1))) issue(g + NS - 1, LPS == 4 ? s : s >> 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[t][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += v[i];
    out[blockIdx.x * NT + tid] = sum;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int WAVES, int VALU, int NS = 5, int MODE = 0>
static void run(const half_t* w, float* out, long long* cyc, int blocks, int nslabs, int stagger = 0)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(walk<WAVES, VALU, NS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, NS * SLAB);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((walk<WAVES, VALU, NS, MODE>), dim3(blocks), dim3(WAVES * 64), NS * SLAB, 0, w, nslabs, out, cyc, stagger);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (long long c : h) s += c;
    printf("%sring %d slots, %d waves per CU, %d VALU per MFMA, stagger %d: %7.0f cycles per 16 KB slab (%d workgroups, %d slabs)\n", MODE == 1 ? "[no MFMA] " : MODE == 2 ? "[no DMA] " : MODE == 3 ? "[head prefetch] " : MODE == 4 ? "[no DMA, no barrier] " : MODE == 5 ? "[no DMA, one fragment read per tile and slab] " : "", NS, WAVES, VALU,
           stagger, s / blocks / nslabs, blocks, nslabs);
}

int main()
{
    const int blocks = 255, nslabs = 126;
    half_t* w; float* out; long long* cyc;
    hipMalloc(&w, 4 * C * C * sizeof(half_t) + (1 << 20));
    hipMemset(w, 0, 4 * C * C * sizeof(half_t) + (1 << 20));
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    run<4, 0>(w, out, cyc, blocks, nslabs);
    run<4, 4>(w, out, cyc, blocks, nslabs);
    run<4, 8>(w, out, cyc, blocks, nslabs);
    run<8, 0>(w, out, cyc, blocks, nslabs);
    run<8, 4>(w, out, cyc, blocks, nslabs);
    run<8, 8>(w, out, cyc, blocks, nslabs);
    // is it the CU or something the CUs share? fewer workgroups; the same number with staggered starts
    for (int nb : {1, 8, 32, 64, 128}) run<4, 0>(w, out, cyc, nb, nslabs);
    for (int st : {1, 3, 7, 24}) run<4, 0>(w, out, cyc, blocks, nslabs, st);
    run<4, 8>(w, out, cyc, blocks, nslabs, 7);
    // is it the depth of the prefetch (bytes in flight / latency)?
    run<4, 0, 4>(w, out, cyc, blocks, nslabs);
    run<4, 0, 6>(w, out, cyc, blocks, nslabs);
    run<4, 0, 8>(w, out, cyc, blocks, nslabs);
    run<4, 0, 9>(w, out, cyc, blocks, nslabs);
    run<4, 8, 8>(w, out, cyc, blocks, nslabs);
    run<8, 8, 8>(w, out, cyc, blocks, nslabs);
    // which side is it? the stream alone, the matrix work alone
    run<4, 0, 5, 1>(w, out, cyc, blocks, nslabs);
    run<8, 0, 5, 1>(w, out, cyc, blocks, nslabs);
    run<4, 0, 5, 2>(w, out, cyc, blocks, nslabs);
    run<4, 8, 5, 2>(w, out, cyc, blocks, nslabs);
    run<8, 8, 5, 2>(w, out, cyc, blocks, nslabs);
    run<4, 0, 5, 4>(w, out, cyc, blocks, nslabs);
    run<4, 0, 5, 5>(w, out, cyc, blocks, nslabs);
    run<4, 8, 5, 5>(w, out, cyc, blocks, nslabs);
    run<4, 0, 5, 3>(w, out, cyc, blocks, nslabs);
    run<4, 8, 5, 3>(w, out, cyc, blocks, nslabs);
    run<8, 0, 5, 3>(w, out, cyc, blocks, nslabs);
    run<8, 8, 5, 3>(w, out, cyc, blocks, nslabs);
    return 0;
}
