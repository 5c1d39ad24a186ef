// Round 6 probe: what does a launch boundary cost against an in-kernel grid-wide barrier on MI355X?
//
//   (a) N dependent launches of an (almost) empty kernel on one stream, eager and as a hipGraph: time per launch boundary
//   (b) ONE persistent launch of G workgroups (1 per CU, 512 threads, 150 KB of LDS each like the block kernel) that
//       runs N phases separated by a grid barrier built from agent-scope atomics (release: L2 write-back, acquire: L2
//       invalidate - what data handed between workgroups on different XCDs needs); every phase each workgroup writes
//       `bytes` of data another workgroup reads in the next phase (checked)
//   (c) the same through hipLaunchCooperativeKernel (co-residency guaranteed by the runtime), eager and captured
//   (d) a ticket-ordered variant: work items taken from an atomic counter, dependencies only on EARLIER tickets
//       (deadlock-free without co-residency)
// Every spin is BOUNDED (gives up after ~20 ms and reports): the probe cannot hang the box.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/grid_sync.hip -o tools/_bin/grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); } } while (0)

__global__ void __launch_bounds__(512) tiny_kernel(int* p, int v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v;
}

// sense-free counting barrier: phase k waits until the counter reaches (k + 1) * G
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int* fail)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000) { ok = false; atomicAdd(fail, 1); break; }   // 20 ms at 100 MHz
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

// variant 1: relaxed polling (no L2 invalidate per poll), ONE acquire fence behind the loop
// variant 2: as 1, without the release's L2 write-back (relaxed add): the cost of the atomics alone (NOT a correct barrier)
// variant 3: two levels - the workgroups of an XCD count on the XCD's own counter, the last one adds to the global counter
__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

template <int VAR>
__device__ __forceinline__ bool grid_barrier_v(unsigned* ctr, unsigned* xctr, unsigned phase, unsigned G, int* fail)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned target = (phase + 1) * G;
        if (VAR == 1) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (VAR == 2) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // 32 workgroups per XCD (G = 256, round-robin over 8 XCDs)
            const unsigned x = xcc_id();
            const unsigned old = __hip_atomic_fetch_add(xctr + 32 * x, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if ((old + 1) % (G / 8) == 0) __hip_atomic_fetch_add(ctr, G / 8, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000) { ok = false; atomicAdd(fail, 1); break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

template <int VAR>
__global__ void __launch_bounds__(512, 1) persistent_v_kernel(unsigned* ctr, unsigned* xctr, int* fail, int* buf, int per_wg_ints, int phases, long long* cycles)
{
    extern __shared__ char lds[];
    const int G = gridDim.x, b = blockIdx.x;
    lds[threadIdx.x] = 0;
    int bad = 0;
    for (int ph = 0; ph < phases; ++ph) {
        int* mine = buf + (static_cast<size_t>(ph & 1) * G + b) * per_wg_ints;
        for (int i = threadIdx.x; i < per_wg_ints; i += 512) mine[i] = ph * 1000 + b;
        if (!grid_barrier_v<VAR>(ctr, xctr, ph, G, fail)) break;
        const int nb = (b + 97) % G;
        const int* theirs = buf + (static_cast<size_t>(ph & 1) * G + nb) * per_wg_ints;
        for (int i = threadIdx.x; i < per_wg_ints; i += 512) bad += theirs[i] != ph * 1000 + nb;
    }
    if (bad) atomicAdd(fail + 1, bad);
}

// phases: workgroup b writes buf[phase & 1][b][...] = phase * 1000 + b, then barrier, then reads its neighbour's (b + 97) % G
__global__ void __launch_bounds__(512, 1) persistent_kernel(unsigned* ctr, int* fail, int* buf, int per_wg_ints, int phases, long long* cycles)
{
    extern __shared__ char lds[];
    const int G = gridDim.x, b = blockIdx.x;
    lds[threadIdx.x] = 0;
    int bad = 0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int ph = 0; ph < phases; ++ph) {
        int* mine = buf + (static_cast<size_t>(ph & 1) * G + b) * per_wg_ints;
        for (int i = threadIdx.x; i < per_wg_ints; i += 512) mine[i] = ph * 1000 + b;
        if (!grid_barrier(ctr, static_cast<unsigned>(ph + 1) * G, fail)) break;
        const int nb = (b + 97) % G;
        const int* theirs = buf + (static_cast<size_t>(ph & 1) * G + nb) * per_wg_ints;
        for (int i = threadIdx.x; i < per_wg_ints; i += 512) bad += theirs[i] != ph * 1000 + nb;
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(fail + 1, bad);
    if (threadIdx.x == 0) cycles[b] = t1 - t0;
}

__global__ void __launch_bounds__(512, 1) coop_kernel(int* fail, int* buf, int per_wg_ints, int phases, long long* cycles)
{
    extern __shared__ char lds[];
    namespace cg = cooperative_groups;
    cg::grid_group grid = cg::this_grid();
    const int G = gridDim.x, b = blockIdx.x;
    lds[threadIdx.x] = 0;
    int bad = 0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int ph = 0; ph < phases; ++ph) {
        int* mine = buf + (static_cast<size_t>(ph & 1) * G + b) * per_wg_ints;
        for (int i = threadIdx.x; i < per_wg_ints; i += 512) mine[i] = ph * 1000 + b;
        grid.sync();
        const int nb = (b + 97) % G;
        const int* theirs = buf + (static_cast<size_t>(ph & 1) * G + nb) * per_wg_ints;
        for (int i = threadIdx.x; i < per_wg_ints; i += 512) bad += theirs[i] != ph * 1000 + nb;
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(fail + 1, bad);
    if (threadIdx.x == 0) cycles[b] = t1 - t0;
}

// ticket order: item = phase * T + tile; item (ph, t) needs items (ph - 1, t - 2 .. t + 2) done. flags[ph][t] set with release.
__global__ void __launch_bounds__(512, 1) ticket_kernel(unsigned* ticket, unsigned* flags, int* fail, int* buf, int per_tile_ints, int phases, int T, long long* cycles)
{
    extern __shared__ char lds[];
    __shared__ unsigned s_item;
    lds[threadIdx.x] = 0;
    int bad = 0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned total = static_cast<unsigned>(phases) * T;
    for (;;) {
        if (threadIdx.x == 0) s_item = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned item = s_item;
        __syncthreads();
        if (item >= total) break;
        const int ph = item / T, t = item % T;
        bool ok = true;
        if (ph > 0) {
            if (threadIdx.x < 5) {
                const int d = t + static_cast<int>(threadIdx.x) - 2;
                if (d >= 0 && d < T) {
                    const long long w0 = __builtin_amdgcn_s_memrealtime();
                    while (__hip_atomic_load(flags + (ph - 1) * T + d, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        __builtin_amdgcn_s_sleep(1);
                        if (__builtin_amdgcn_s_memrealtime() - w0 > 2000000) { atomicAdd(fail, 1); break; }
                    }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int dd = -2; dd <= 2; ++dd) {
                const int d = t + dd;
                if (d < 0 || d >= T) continue;
                const int* theirs = buf + (static_cast<size_t>((ph - 1) & 1) * T + d) * per_tile_ints;
                for (int i = threadIdx.x; i < per_tile_ints; i += 512 * 8) bad += theirs[i] != (ph - 1) * 1000 + d;
            }
        }
        (void)ok;
        int* mine = buf + (static_cast<size_t>(ph & 1) * T + t) * per_tile_ints;
        for (int i = threadIdx.x; i < per_tile_ints; i += 512) mine[i] = ph * 1000 + t;
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + ph * T + t, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (bad) atomicAdd(fail + 1, bad);
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 40;
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int coop = 0;
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    printf("CUs %d, cooperative launch attribute %d, %d phases / launches\n", cus, coop, N);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int* d_p; CK(hipMalloc(&d_p, 64));
    unsigned* d_ctr; CK(hipMalloc(&d_ctr, 64));
    int* d_fail; CK(hipMalloc(&d_fail, 64));
    long long* d_cyc; CK(hipMalloc(&d_cyc, 8 * 1024));
    const int LDS = 150 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ticket_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    // ---- (a) launch boundaries
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipStreamSynchronize(st));
        const double h0 = now_us();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(cus), dim3(512), 0, st, d_p, i);
        CK(hipEventRecord(e1, st));
        const double h1 = now_us();
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("(a) eager: %.2f us per launch on the GPU timeline, %.2f us of host time per launch\n", ms * 1000 / N, (h1 - h0) / N);
    }
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(cus), dim3(512), 0, st, d_p, i);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("(a) graph: %.2f us per launch\n", ms * 1000 / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }

    // ---- (b) persistent kernel with atomic grid barriers
    for (int kb : {0, 4, 64, 200}) {
        const int per = kb * 1024 / 4 > 0 ? kb * 1024 / 4 : 16;
        int* d_buf; CK(hipMalloc(&d_buf, static_cast<size_t>(2) * cus * per * 4));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(d_ctr, 0, 64, st)); CK(hipMemsetAsync(d_fail, 0, 64, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent_kernel, dim3(cus), dim3(512), LDS, st, d_ctr, d_fail, d_buf, per, N, d_cyc);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            int fail[2]; CK(hipMemcpy(fail, d_fail, 8, hipMemcpyDeviceToHost));
            if (rep == 2) printf("(b) persistent, %3d KB written per workgroup and phase: %.2f us per phase (launch %.1f us), barrier timeouts %d, stale reads %d\n", kb, ms * 1000 / N, ms * 1000, fail[0], fail[1]);
        }
        CK(hipFree(d_buf));
    }

    // ---- (b') barrier variants
    {
        unsigned* d_x; CK(hipMalloc(&d_x, 8 * 32 * 4));
        auto run_v = [&](auto kern, const char* name, int kb) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            const int per = kb * 1024 / 4 > 0 ? kb * 1024 / 4 : 16;
            int* d_buf; CK(hipMalloc(&d_buf, static_cast<size_t>(2) * cus * per * 4));
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(d_ctr, 0, 64, st)); CK(hipMemsetAsync(d_fail, 0, 64, st)); CK(hipMemsetAsync(d_x, 0, 8 * 32 * 4, st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(kern, dim3(cus), dim3(512), LDS, st, d_ctr, d_x, d_fail, d_buf, per, N, d_cyc);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                int fail[2]; CK(hipMemcpy(fail, d_fail, 8, hipMemcpyDeviceToHost));
                if (rep == 2) printf("(b') %s, %3d KB per workgroup and phase: %.2f us per phase, barrier timeouts %d, stale reads %d\n", name, kb, ms * 1000 / N, fail[0], fail[1]);
            }
            CK(hipFree(d_buf));
        };
        for (int kb : {0, 64}) {
            run_v(persistent_v_kernel<1>, "relaxed polling + one acquire fence", kb);
            run_v(persistent_v_kernel<2>, "relaxed add, relaxed polling (atomics alone; stale reads expected)", kb);
            run_v(persistent_v_kernel<3>, "two levels (per-XCD counters), relaxed polling", kb);
        }
        CK(hipFree(d_x));
    }

    // ---- (c) cooperative launch
    if (coop) {
        const int per = 4 * 1024 / 4;
        int* d_buf; CK(hipMalloc(&d_buf, static_cast<size_t>(2) * cus * per * 4));
        int n = N, perv = per;
        void* args[] = {&d_fail, &d_buf, &perv, &n, &d_cyc};
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(d_fail, 0, 64, st));
            const double h0 = now_us();
            CK(hipEventRecord(e0, st));
            CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(coop_kernel), dim3(cus), dim3(512), args, LDS, st));
            CK(hipEventRecord(e1, st));
            const double h1 = now_us();
            CK(hipStreamSynchronize(st));
            const double h2 = now_us();
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            int fail[2]; CK(hipMemcpy(fail, d_fail, 8, hipMemcpyDeviceToHost));
            if (rep == 2) printf("(c) cooperative: %.2f us per phase (launch %.1f us on the GPU timeline, host %.1f us to submit, %.1f us to completion), stale reads %d\n", ms * 1000 / N, ms * 1000, h1 - h0, h2 - h0, fail[1]);
        }
        // ten short cooperative launches back to back: the per-launch cost of the cooperative queue
        n = 1;
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(coop_kernel), dim3(cus), dim3(512), args, LDS, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(c) ten one-phase cooperative launches back to back: %.2f us each\n", ms * 100);
        // mixed with ordinary launches (does the cooperative queue cost a cross-queue dependency each way?)
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) {
            hipLaunchKernelGGL(tiny_kernel, dim3(cus), dim3(512), 0, st, d_p, i);
            CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(coop_kernel), dim3(cus), dim3(512), args, LDS, st));
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(c) ten [ordinary + cooperative] pairs: %.2f us per pair\n", ms * 100);
        // captured?
        n = N;
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        hipError_t el = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(coop_kernel), dim3(cus), dim3(512), args, LDS, st);
        hipError_t ee = hipStreamEndCapture(st, &g);
        printf("(c) capture: begin %s, launch %s, end %s\n", hipGetErrorString(e), hipGetErrorString(el), hipGetErrorString(ee));
        (void)hipGetLastError();
        if (ee == hipSuccess && g != nullptr && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(d_fail, 0, 64, st));
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&ms, e0, e1));
                int fail[2]; CK(hipMemcpy(fail, d_fail, 8, hipMemcpyDeviceToHost));
                if (rep == 2) printf("(c) cooperative inside a graph: %.2f us per phase, stale reads %d\n", ms * 1000 / N, fail[1]);
            }
        } else {
            printf("(c) a cooperative launch cannot be instantiated from a capture here\n");
            (void)hipGetLastError();
        }
        CK(hipFree(d_buf));
    }

    // ---- (d) ticket order
    for (int T : {cus, 2 * cus}) {
        const int per = 64 * 1024 / 4;
        int* d_buf; CK(hipMalloc(&d_buf, static_cast<size_t>(2) * T * per * 4));
        unsigned* d_flags; CK(hipMalloc(&d_flags, static_cast<size_t>(N) * T * 4));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(d_ctr, 0, 64, st)); CK(hipMemsetAsync(d_fail, 0, 64, st));
            CK(hipMemsetAsync(d_flags, 0, static_cast<size_t>(N) * T * 4, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(ticket_kernel, dim3(cus), dim3(512), LDS, st, d_ctr, d_flags, d_fail, d_buf, per, N, T, d_cyc);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            int fail[2]; CK(hipMemcpy(fail, d_fail, 8, hipMemcpyDeviceToHost));
            if (rep == 2) printf("(d) ticket order, %d tiles per phase, 64 KB per tile: %.2f us per phase, flag timeouts %d, stale reads %d\n", T, ms * 1000 / N, fail[0], fail[1]);
        }
        // half the grid: no co-residency assumption - it must still finish
        CK(hipMemsetAsync(d_ctr, 0, 64, st)); CK(hipMemsetAsync(d_fail, 0, 64, st));
        CK(hipMemsetAsync(d_flags, 0, static_cast<size_t>(N) * T * 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(ticket_kernel, dim3(cus / 2), dim3(512), LDS, st, d_ctr, d_flags, d_fail, d_buf, per, N, T, d_cyc);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        int fail[2]; CK(hipMemcpy(fail, d_fail, 8, hipMemcpyDeviceToHost));
        printf("(d) the same on HALF the workgroups: %.2f us per phase, flag timeouts %d, stale reads %d\n", ms * 1000 / N, fail[0], fail[1]);
        CK(hipFree(d_buf)); CK(hipFree(d_flags));
    }
    return 0;
}
