// Machinery shared by the picture codecs (intra DMCI, inter LD / HT-S / HT-L): the codec's own
// compute stream, per-stage hipGraph slots, the entropy-coding worker thread with its
// high-priority transfer stream, pinned staging buffers and the rANS coders.
//
// Reference counterpart: the run()/m_gexec_* CUDA-graph helpers, worker thread and pinned size
// pointer that every *_proxy.cpp re-implements (e.g. dmc_ld_proxy.cpp:373-405, 809-900).
#pragma once

#include "codec/modules.h"
#include "rans/rans_coder.h"

#include <initializer_list>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

namespace dcvc {

constexpr int kMinSymbolsPerStream = 32768;   // def_const.h:18
constexpr int kQpNum = 64;                    // common_model.py:135-136

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// dmc_common.cpp:31-35
inline int ec_parallel_for(int symbols)
{
    const int p = symbols / kMinSymbolsPerStream;
    return p < 1 ? 1 : (p > kMaxEcParallel ? kMaxEcParallel : p);
}

// Pinned host buffer that only ever grows.
template <typename T>
class Pinned {
public:
    Pinned() = default;
    ~Pinned() { release(); }
    Pinned(const Pinned&) = delete;
    Pinned& operator=(const Pinned&) = delete;
    void reserve(size_t count)
    {
        if (count <= m_cap) return;
        release();
        hip_check(hipHostMalloc(reinterpret_cast<void**>(&m_p), count * sizeof(T), hipHostMallocDefault),
                  "hipHostMalloc");
        m_cap = count;
    }
    T* get() const { return m_p; }
    T& operator[](size_t i) const { return m_p[i]; }

private:
    void release()
    {
        if (m_p) (void)hipHostFree(m_p);
        m_p = nullptr;
        m_cap = 0;
    }
    T* m_p = nullptr;
    size_t m_cap = 0;
};

class CodecBase {
public:
    void set_use_graphs(bool on) { m_use_graphs = on; }
    const std::vector<uint8_t>& stream_bytes() const { return m_enc.stream(); }

protected:
    CodecBase();
    ~CodecBase();
    CodecBase(const CodecBase&) = delete;
    CodecBase& operator=(const CodecBase&) = delete;

    // All codec work runs on the codec's own non-blocking stream, ordered after the caller's
    // stream on entry and before it on exit: graph capture then works whatever stream the caller
    // is on (the legacy default stream cannot capture - the reference harness has to switch
    // streams for that reason, test_video.py:422-425).
    hipStream_t enter(hipStream_t user);
    void leave(hipStream_t user);

    // Stage = a fixed launch sequence. First call runs eagerly (lazy one-time initialisation
    // happens there), the second is captured, later ones replay the instantiated graph.
    template <typename F>
    void run_stage(int key, hipStream_t st, F&& fn);
    // Declares a pointer that is baked into the capture of `key`; a change drops the graph.
    void bind_stage_arg(int key, const void* arg);
    void clear_graphs();
    // Waits for everything the codec has in flight; derived destructors call it before their
    // device buffers go away.
    void quiesce();

    // Entropy worker: one job at a time. submit() records `after` on the compute stream and makes
    // the io stream wait for it (before the caller may start capturing the next stage: HIP refuses
    // cross-stream waits on a capturing stream).
    void submit(hipStream_t st, std::function<void()> job);
    void wait_job();               // rethrows a worker failure as std::runtime_error

    void load_cdf_tables(const ParamStore& ps);   // both coders, z = table 0, y = table 1
    const half_t* upload_qp_table(const ParamStore& ps, DeviceArena& mem, const char* name, int ch);
    // per-qp scale vector -> fixed device slot, so that one graph serves all 64 qps
    // (ONE launch for all the vectors of a codec: the 3-4 separate 1 KB device copies of round 1 showed up as
    // 8 copy launches per coded picture in the kernel trace)
    struct QpRow {
        half_t* dst;
        const half_t* table;
        int ch;
    };
    static void copy_qp_rows(std::initializer_list<QpRow> rows, int qp, hipStream_t st);

    RansEncoder m_enc;
    RansDecoder m_dec;
    hipStream_t m_io_stream = nullptr;    // D2H / H2D of symbols, high priority
    hipStream_t m_cs = nullptr;           // the codec's compute stream
    hipStream_t m_join = nullptr;    // blocking stream that joins our results to the legacy null stream
    bool m_use_graphs = true;

private:
    void worker_loop();

    struct GraphSlot {
        hipGraphExec_t exec = nullptr;
        bool warmed = false;
        const void* arg = nullptr;
    };
    std::map<int, GraphSlot> m_graphs;
    hipEvent_t m_ev_job = nullptr, m_ev_in = nullptr, m_ev_out = nullptr;
    std::thread m_worker;
    std::mutex m_mu;
    std::condition_variable m_cv_work, m_cv_done;
    std::function<void()> m_job;
    bool m_pending = false, m_done = true, m_stop = false;
    std::string m_worker_error;
};

template <typename F>
void CodecBase::run_stage(int key, hipStream_t st, F&& fn)
{
    if (!m_use_graphs) {
        fn();
        return;
    }
    GraphSlot& slot = m_graphs[key];
    if (!slot.warmed) {
        fn();
        slot.warmed = true;
        return;
    }
    if (!slot.exec) {
        hipGraph_t graph = nullptr;
        hip_check(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
        try {
            fn();
        } catch (...) {
            (void)hipStreamEndCapture(st, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        hip_check(hipStreamEndCapture(st, &graph), "hipStreamEndCapture");
        const hipError_t e = hipGraphInstantiate(&slot.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        hip_check(e, "hipGraphInstantiate");
    }
    hip_check(hipGraphLaunch(slot.exec, st), "hipGraphLaunch");
}

}  // namespace dcvc
