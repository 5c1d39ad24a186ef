// core_bench - A/B timing of the block kernels of several builds of libdcvc_amd.so in ONE process, without
// Python (a fresh GPU box spends 1-2 minutes in `import torch`; this starts at once).
//
//   core_bench [-p pixels] [-r rounds] [-n launches] lib_a.so [lib_b.so ...]
//
// Every library is dlopen()ed privately; per round and library: n launches of dcvc_dcb_core (with and without the
// next block's dc.0) and of the 4-launch conv1x1 sequence it replaces, bracketed by HIP events on one stream. Rounds are
// interleaved over the libraries (drift, DVFS). Operands: uniform random fp16 (never zeros: a zero-filled matrix
// core clocks 15-20 % higher). Prints per library the median microseconds, TFLOP/s, a checksum of y, and the in-kernel
// timeline of dcb_core (shader-clock stamps per phase, median over workgroups).
//
// build: hipcc -O2 --offload-arch=gfx950 tools/probes/core_bench.hip -o tools/_bin/core_bench -ldl
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

typedef int (*core_fn)(const void*, int, const void*, int, const void*, const void*, const void*, const void*, const void*,
                       const void*, const void*, const void*, const void*, const void*, void*, int, void*, int, int, int, int, void*);
typedef int (*ns_fn)(const void*, int, const void*, int, const void*, const void*, const void*, const void*, const void*,
                     const void*, const void*, const void*, const void*, const void*, void*, int, void*, int, int, int, int, int, void*);
typedef int (*nspack_fn)(const void*, const void*, const void*, const void*, int, int, void*, void**);
typedef int (*nspacked_fn)(const void*, const void*, int, const void*, int, const void*, const void*, const void*, const void*, const void*,
                           const void*, void*, int, void*, int, int, int, int, void*);
typedef int (*conv_fn)(const void*, int, const void*, const void*, const void*, int, const void*, int, const void*, const void*,
                       void*, int, int, int, int, int, void*);
typedef int (*dw_fn)(const void*, int, const void*, void*, int, int, int, int, void*);
typedef int (*tl_fn)(void*);
typedef int (*dwhook_fn)(const void*, const void*, int);
typedef const char* (*err_fn)(void);

struct Lib {
    std::string path;
    void* h = nullptr;
    core_fn core = nullptr;
    ns_fn ns = nullptr;            // dcvc_dcb_nsplit (+ the inner width), round 3
    nspack_fn ns_pack = nullptr;   // round 4: dcvc_dcb_nsplit packs per call; the handle form is what gets timed
    nspacked_fn ns_packed = nullptr;
    bool has_core = false;         // round 2's dcb_core is in the library (since round 4 only in -DDCVC_WITH_DCB_CORE builds) and the shape is its
    void* ns_handle = nullptr;
    conv_fn conv = nullptr;
    dw_fn dw = nullptr;
    tl_fn tl = nullptr;
    tl_fn ns_tl = nullptr;
    dwhook_fn dw_hook = nullptr;   // round 6: dcvc_dcb_nsplit_dw_hook
    err_fn err = nullptr;
    std::vector<float> us_core, us_core_next, us_seq, us_dw, us_ns, us_ns_next;
};

static uint16_t f2h(float f)
{
    _Float16 h = static_cast<_Float16>(f);
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

static void* device_random(size_t n, float scale, std::mt19937& rng)
{
    std::vector<uint16_t> host(n);
    std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : host) v = f2h(d(rng));
    void* p = nullptr;
    OK(hipMalloc(&p, n * 2));
    OK(hipMemcpy(p, host.data(), n * 2, hipMemcpyHostToDevice));
    return p;
}

static float median(std::vector<float> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.f : v[v.size() / 2];
}

int main(int argc, char** argv)
{
    int P = 32640, rounds = 5, n = 20, H = 136, W = 240, C = 384, CI = 0;
    bool dw_inside = false;        // -w: the block launches take their depthwise conv inside (shapes that have the variant)
    std::vector<Lib> libs;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-p") && i + 1 < argc) { P = atoi(argv[++i]); H = 1; W = P; }
        else if (!strcmp(argv[i], "-r") && i + 1 < argc) rounds = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-n") && i + 1 < argc) n = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-c") && i + 1 < argc) C = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-i") && i + 1 < argc) CI = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-w")) dw_inside = true;
        else if (!strcmp(argv[i], "-g") && i + 2 < argc) { H = atoi(argv[++i]); W = atoi(argv[++i]); P = H * W; }
        else { Lib l; l.path = argv[i]; libs.push_back(l); }
    }
    if (CI == 0) CI = C;
    const bool core_shape = C == 384 && CI == 384;
    if (libs.empty()) { fprintf(stderr, "usage: core_bench [-p pixels] [-r rounds] [-n launches] [-c block width] [-i inner width] lib.so ...\n"); return 2; }
    for (auto& l : libs) {
        l.h = dlopen(l.path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", l.path.c_str(), dlerror()); return 1; }
        l.core = reinterpret_cast<core_fn>(dlsym(l.h, "dcvc_dcb_core"));
        l.conv = reinterpret_cast<conv_fn>(dlsym(l.h, "dcvc_conv1x1"));
        l.ns = reinterpret_cast<ns_fn>(dlsym(l.h, "dcvc_dcb_nsplit"));
        l.ns_pack = reinterpret_cast<nspack_fn>(dlsym(l.h, "dcvc_dcb_nsplit_pack"));
        l.ns_packed = reinterpret_cast<nspacked_fn>(dlsym(l.h, "dcvc_dcb_nsplit_packed"));
        l.dw = reinterpret_cast<dw_fn>(dlsym(l.h, "dcvc_dwconv3x3"));
        l.tl = reinterpret_cast<tl_fn>(dlsym(l.h, "dcvc_dcb_core_timeline_buffer"));
        l.ns_tl = reinterpret_cast<tl_fn>(dlsym(l.h, "dcvc_dcb_nsplit_timeline_buffer"));
        l.dw_hook = reinterpret_cast<dwhook_fn>(dlsym(l.h, "dcvc_dcb_nsplit_dw_hook"));
        l.err = reinterpret_cast<err_fn>(dlsym(l.h, "dcvc_last_error"));
        if (!l.conv || !l.err) { fprintf(stderr, "%s: missing symbols\n", l.path.c_str()); return 1; }      // (dcvc_dcb_core: libraries up to round 4 only)
    }
    std::mt19937 rng(1234);
    const float ws = 1.7f / sqrtf(static_cast<float>(C));       // keeps activations O(1) through the block
    void* x = device_random(static_cast<size_t>(P) * C, 1.0f, rng);
    const float wsi = 1.7f / sqrtf(static_cast<float>(CI));
    void* t2 = device_random(static_cast<size_t>(P) * CI, 1.0f, rng);
    void* w3 = device_random(static_cast<size_t>(C) * CI, wsi, rng);
    void* w2 = device_random(static_cast<size_t>(C) * CI, wsi, rng);
    void* w1 = device_random(static_cast<size_t>(CI) * C, ws, rng);
    void* w0 = device_random(static_cast<size_t>(4) * CI * C, ws, rng);
    void* wd = device_random(static_cast<size_t>(9) * CI, 0.3f, rng);
    void* b3 = device_random(C, 0.5f, rng);
    void* b2 = device_random(C, 0.5f, rng);
    void* b1 = device_random(CI, 0.5f, rng);
    void* b0 = device_random(4 * CI, 0.5f, rng);
    void *y, *t1, *y1, *t;
    for (void** p : {&y, &t1, &y1, &t}) { OK(hipMalloc(p, static_cast<size_t>(P) * C * 2)); OK(hipMemset(*p, 0, static_cast<size_t>(P) * C * 2)); }
    hipStream_t st;
    OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    auto chk = [&](const Lib& l, int rc, const char* what) { if (rc < 0) { fprintf(stderr, "%s: %s: %s\n", l.path.c_str(), what, l.err()); exit(1); } };
    auto core = [&](Lib& l, bool next) {
        chk(l, l.core(t2, C, x, C, w3, b3, w0, b0, w2, b2, nullptr, nullptr, next ? w1 : nullptr, next ? b1 : nullptr, next ? t1 : nullptr, C,
                      y, C, P, C, 0, st), "dcb_core");
    };
    for (auto& l : libs) {
        // a library built without dcb_core answers the call with an error: that build's reference result is the launch sequence
        l.has_core = core_shape && l.core != nullptr && l.core(t2, C, x, C, w3, b3, w0, b0, w2, b2, nullptr, nullptr, nullptr, nullptr, nullptr, C, y, C, P, C, 0, st) >= 0;
    }
    OK(hipStreamSynchronize(st));
    if (dw_inside) {
        for (auto& l : libs) {
            if (!l.dw_hook || H <= 1) { fprintf(stderr, "%s: -w needs dcvc_dcb_nsplit_dw_hook and a picture (no -p)\n", l.path.c_str()); return 1; }
            chk(l, l.dw_hook(t2, wd, W), "dcb_nsplit_dw_hook");
            printf("(-w: dcb_nsplit runs with its depthwise conv inside, on t2 as dc.0's output: its results differ from the reference line's by construction)\n");
        }
    }
    auto nsplit = [&](Lib& l, bool next) {
        if (l.ns_pack && l.ns_packed) {
            if (!l.ns_handle) chk(l, l.ns_pack(w3, w0, w2, w1, C, CI, st, &l.ns_handle), "dcb_nsplit_pack");
            chk(l, l.ns_packed(l.ns_handle, t2, CI, x, C, b3, b0, b2, nullptr, nullptr, next ? b1 : nullptr, next ? t1 : nullptr, CI,
                               y, C, P, 0, next ? 1 : 0, st), "dcb_nsplit_packed");
            return;
        }
        chk(l, l.ns(t2, CI, x, C, w3, b3, w0, b0, w2, b2, nullptr, nullptr, next ? w1 : nullptr, next ? b1 : nullptr, next ? t1 : nullptr, CI,
                    y, C, P, C, CI, 0, st), "dcb_nsplit");
    };
    auto seq = [&](Lib& l) {
        chk(l, l.conv(t2, CI, w3, b3, x, C, nullptr, 0, nullptr, nullptr, y1, C, P, CI, C, 0, st), "conv");
        chk(l, l.conv(y1, C, w0, b0, nullptr, 0, nullptr, 0, nullptr, nullptr, t, CI, P, C, 4 * CI, 3, st), "conv");
        chk(l, l.conv(t, CI, w2, b2, y1, C, nullptr, 0, nullptr, nullptr, y, C, P, CI, C, 0, st), "conv");
        chk(l, l.conv(y, C, w1, b1, nullptr, 0, nullptr, 0, nullptr, nullptr, t1, CI, P, C, CI, 1, st), "conv");
    };
    auto timed = [&](auto&& fn) {
        for (int i = 0; i < 3; ++i) fn();
        OK(hipStreamSynchronize(st));
        OK(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) fn();
        OK(hipEventRecord(e1, st));
        OK(hipEventSynchronize(e1));
        float ms = 0;
        OK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / n;
    };
    for (int r = 0; r < rounds; ++r) {
        for (auto& l : libs) {
            if (l.has_core) {
                l.us_core_next.push_back(timed([&] { core(l, true); }));
                l.us_core.push_back(timed([&] { core(l, false); }));
            }
            l.us_seq.push_back(timed([&] { seq(l); }));
            if (l.ns) {
                l.us_ns_next.push_back(timed([&] { nsplit(l, true); }));
                l.us_ns.push_back(timed([&] { nsplit(l, false); }));
            }
            if (l.dw && H > 1) l.us_dw.push_back(timed([&] { chk(l, l.dw(t2, CI, wd, t, CI, H, W, CI, st), "dwconv"); }));
        }
    }
    const double flop = 2.0 * P * 7 * C * CI;
    std::vector<uint16_t> hy(static_cast<size_t>(P) * C);
    for (auto& l : libs) {
        if (l.has_core) core(l, true); else seq(l);      // the reference result: dcb_core where it exists, else the four launches
        OK(hipStreamSynchronize(st));
        OK(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost));
        uint64_t sum = 0;
        for (size_t i = 0; i < hy.size(); ++i) sum = sum * 1315423911u + hy[i];
        OK(hipMemcpy(hy.data(), t1, hy.size() * 2, hipMemcpyDeviceToHost));
        uint64_t sum1 = 0;
        for (size_t i = 0; i < hy.size(); ++i) sum1 = sum1 * 1315423911u + hy[i];
        const float a = median(l.us_core_next), b = median(l.us_core), c = median(l.us_seq);
        printf("%s\n  dcb_core + next dc.0 %7.1f us %6.0f TFLOP/s | dcb_core %7.1f us %6.0f TFLOP/s | 4 conv1x1 %7.1f us %6.0f TFLOP/s | dw3x3 %6.1f us"
               " | y %016llx t1 %016llx\n", l.path.c_str(), a, flop / a / 1e6, b, flop * 6 / 7 / b / 1e6, c, flop / c / 1e6, median(l.us_dw),
               static_cast<unsigned long long>(sum), static_cast<unsigned long long>(sum1));
        if (l.ns) {
            OK(hipMemset(y, 0, hy.size() * 2));
            OK(hipMemset(t1, 0, hy.size() * 2));
            OK(hipDeviceSynchronize());        // the memsets run on the null stream, which `st` (non-blocking) does not wait for
            nsplit(l, true);
            OK(hipStreamSynchronize(st));
            OK(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost));
            uint64_t s2 = 0;
            for (size_t i = 0; i < hy.size(); ++i) s2 = s2 * 1315423911u + hy[i];
            OK(hipMemcpy(hy.data(), t1, hy.size() * 2, hipMemcpyDeviceToHost));
            uint64_t s3 = 0;
            for (size_t i = 0; i < hy.size(); ++i) s3 = s3 * 1315423911u + hy[i];
            const float an = median(l.us_ns_next), bn = median(l.us_ns);
            printf("  dcb_nsplit + next dc.0 %7.1f us %6.0f TFLOP/s | dcb_nsplit %7.1f us %6.0f TFLOP/s | y %016llx t1 %016llx %s\n", an, flop / an / 1e6,
                   bn, flop * 6 / 7 / bn / 1e6, static_cast<unsigned long long>(s2), static_cast<unsigned long long>(s3),
                   (s2 == sum && s3 == sum1) ? "(= reference result)" : "(DIFFERS from the reference result)");
        }
        if (l.ns && l.ns_tl) {
            long long* tl = nullptr;
            const size_t rows = 2048;
            OK(hipMalloc(&tl, rows * 32 * 8));
            OK(hipMemset(tl, 0, rows * 32 * 8));
            OK(hipDeviceSynchronize());
            chk(l, l.ns_tl(tl), "timeline");
            nsplit(l, true);
            OK(hipStreamSynchronize(st));
            chk(l, l.ns_tl(nullptr), "timeline");
            std::vector<long long> h(rows * 32);
            OK(hipMemcpy(h.data(), tl, rows * 32 * 8, hipMemcpyDeviceToHost));
            OK(hipFree(tl));
            static const char* names[13] = {"prologue", "dc.3 mfma", "x + dc.3 epi + bar", "ffn0.0 mfma", "ffn0.1 mfma | epi 0", "ffn0.2 mfma | epi 1", "epi 2",
                                             "bar + ffn.2 mfma", "epi + bar", "y out", "dc.0 mfma", "epi + bar", "t1n out"};
            printf("  nsplit timeline (median cycles over workgroups):");
            // number of intervals = stamps written by the first workgroup that ran - 1 (4-wave kernel: 10 + ffn.0 passes, see the
            // stamp list in dcb_nsplit_kernel.h; 8-wave kernel: entry | prologue | dc.3 | one per ffn.0 pass | ffn.0 done | ffn.2 MFMAs
            // | ffn.2 | dc.0 MFMAs | dc.0)
            int nstamps = 0;
            for (size_t w = 0; w < rows && nstamps == 0; ++w) {
                if (h[w * 32] == 0) continue;
                for (int i = 1; i < 29 && h[w * 32 + i] != 0; ++i) nstamps = i;
            }
            for (int i = 0; i < nstamps; ++i) {
                std::vector<float> d;
                for (size_t w = 0; w < rows; ++w) if (h[w * 32] != 0 && h[w * 32 + i + 1] != 0) d.push_back(static_cast<float>(h[w * 32 + i + 1] - h[w * 32 + i]));
                if (nstamps == 13) printf(" %s %.0f |", names[i], median(d)); else printf(" [%d] %.0f |", i + 1, median(d));
            }
            std::vector<float> tot, start;
            long long t0 = 0;
            for (size_t w = 0; w < rows; ++w) if (h[w * 32] != 0 && (t0 == 0 || h[w * 32] < t0)) t0 = h[w * 32];
            for (size_t w = 0; w < rows; ++w) if (h[w * 32] != 0) { tot.push_back(static_cast<float>(h[w * 32 + nstamps] - h[w * 32])); start.push_back(static_cast<float>(h[w * 32] - t0)); }
            std::sort(start.begin(), start.end());
            // stamp 31 (round 4) = the workgroup's last instruction: whole-launch cycles per workgroup; with the launch's wall time
            // (us_ns_next) the effective shader clock
            std::vector<float> whole;
            long long tend = 0;
            for (size_t w = 0; w < rows; ++w) if (h[w * 32] != 0 && h[w * 32 + 31] != 0) { whole.push_back(static_cast<float>(h[w * 32 + 31] - h[w * 32])); tend = std::max(tend, h[w * 32 + 31]); }
            printf(" total %.0f | workgroups %zu, start of the median / last workgroup %.0f / %.0f", median(tot), tot.size(),
                   start.empty() ? 0.f : start[start.size() / 2], start.empty() ? 0.f : start.back());
            // (the counters of different XCDs are not aligned: only differences inside one workgroup mean anything)
            // stamps 29 / 30 (round 4): the 100 MHz real-time counter at the workgroup's entry / last instruction
            std::vector<float> ghz, span_us;
            long long rt_first = 0, rt_last = 0;
            for (size_t w = 0; w < rows; ++w) {
                if (h[w * 32] == 0 || h[w * 32 + 30] == 0 || h[w * 32 + 29] == 0) continue;
                const double us = static_cast<double>(h[w * 32 + 30] - h[w * 32 + 29]) / 100.0;
                if (us > 0) { ghz.push_back(static_cast<float>((h[w * 32 + 31] - h[w * 32]) / us / 1e3)); span_us.push_back(static_cast<float>(us)); }
                if (rt_first == 0 || h[w * 32 + 29] < rt_first) rt_first = h[w * 32 + 29];
                rt_last = std::max(rt_last, h[w * 32 + 30]);
            }
            if (!ghz.empty()) printf(" | shader clock inside the launch %.2f GHz (cycles / 100 MHz real-time counter, median workgroup); workgroups run %.1f us (median), "
                                     "first entry to last exit %.1f us of the launch's %.1f us",
                                     median(ghz), median(span_us), static_cast<double>(rt_last - rt_first) / 100.0, median(l.us_ns_next));
            if (!whole.empty()) {
                std::sort(whole.begin(), whole.end());
                printf(" | whole launch, cycles per workgroup: min %.0f, median %.0f, p90 %.0f, max %.0f; max / wall time %.1f us = %.2f GHz",
                       whole.front(), whole[whole.size() / 2], whole[whole.size() * 9 / 10], whole.back(), median(l.us_ns_next),
                       whole.back() / median(l.us_ns_next) / 1e3);
            }
            printf("\n");
        }
        if (l.tl && l.has_core) {
            long long* tl = nullptr;
            const size_t rows = 1024;
            OK(hipMalloc(&tl, rows * 64 * 8));
            OK(hipMemset(tl, 0, rows * 64 * 8));
            OK(hipDeviceSynchronize());
            chk(l, l.tl(tl), "timeline");
            core(l, true);
            OK(hipStreamSynchronize(st));
            chk(l, l.tl(nullptr), "timeline");
            std::vector<long long> h(rows * 64);
            OK(hipMemcpy(h.data(), tl, rows * 64 * 8, hipMemcpyDeviceToHost));
            OK(hipFree(tl));
            static const char* names[11] = {"dc.3 (18 slabs)", "y1 epilogue", "ffn sc0 (12)", "ffn sc1 (15)", "ffn sc2 (15)", "ffn sc3 (15)",
                                             "ffn sc4 (15)", "ffn sc5 (15)", "last pair + ffn.2 (3)", "y epilogue + stores", "next dc.0 (18) + drain"};
            printf("  timeline (median cycles over workgroups):");
            for (int i = 0; i < 11; ++i) {
                std::vector<float> d;
                for (size_t w = 0; w < rows; ++w) if (h[w * 64] != 0) d.push_back(static_cast<float>(h[w * 64 + i + 1] - h[w * 64 + i]));
                printf(" %s %.0f |", names[i], median(d));
            }
            std::vector<float> tot;
            for (size_t w = 0; w < rows; ++w) if (h[w * 64] != 0) tot.push_back(static_cast<float>(h[w * 64 + 11] - h[w * 64]));
            printf(" total %.0f\n", median(tot));
        }
    }
    return 0;
}
