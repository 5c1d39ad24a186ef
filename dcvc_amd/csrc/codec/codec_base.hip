// See codec_base.h.
#include "codec/codec_base.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace dcvc {

CodecBase::CodecBase()
{
    int lo = 0, hi = 0;
    hip_check(hipDeviceGetStreamPriorityRange(&lo, &hi), "hipDeviceGetStreamPriorityRange");
    // DCVC_COMPUTE_PRIORITY = high | low, read when a codec object is created: the priority of its streams. A
    // decoder that shares a GPU with an encoder (bench.py's two-stage loop) is a chain of short kernels and host round
    // trips - at equal priority every one of them queues behind a kernel of the encoder's burst.
    // The transfer stream follows: HIP multiplexes the streams of one priority level onto a few hardware queues, and the
    // transfer stream carries "wait for this object's stage" barriers - an ENCODER's transfer stream at high priority could
    // land on the hardware queue of the DECODER's compute stream and park the decoder's kernels behind the encoder's stage
    // (round 5, profiles/r05_pipeline_order.txt: the same two-stage loop reached 109 or 178 pictures/s depending only on which
    // objects had been created before). An object created without the variable keeps its transfers at high priority.
    int prio = 0, io_prio = hi;
    if (const char* e = getenv("DCVC_COMPUTE_PRIORITY")) {
        const std::string v(e);
        prio = v == "high" ? hi : v == "low" ? lo : 0;
        io_prio = v == "low" ? lo : hi;
    }
    hip_check(hipStreamCreateWithPriority(&m_io_stream, hipStreamNonBlocking, io_prio), "hipStreamCreate(io)");
    hip_check(hipStreamCreateWithPriority(&m_cs, hipStreamNonBlocking, prio), "hipStreamCreate(compute)");
    hip_check(hipEventCreateWithFlags(&m_ev_job, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventCreateWithFlags(&m_ev_in, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventCreateWithFlags(&m_ev_out, hipEventDisableTiming), "hipEventCreate");
    m_worker = std::thread(&CodecBase::worker_loop, this);
}

CodecBase::~CodecBase()
{
    {
        std::lock_guard<std::mutex> lk(m_mu);
        m_stop = true;
    }
    m_cv_work.notify_all();
    if (m_worker.joinable()) m_worker.join();
    clear_graphs();
    if (m_cs) (void)hipStreamSynchronize(m_cs);
    if (m_ev_job) (void)hipEventDestroy(m_ev_job);
    if (m_ev_in) (void)hipEventDestroy(m_ev_in);
    if (m_ev_out) (void)hipEventDestroy(m_ev_out);
    if (m_join) (void)hipStreamDestroy(m_join);
    if (m_cs) (void)hipStreamDestroy(m_cs);
    if (m_io_stream) (void)hipStreamDestroy(m_io_stream);
}

void CodecBase::clear_graphs()
{
    for (auto& kv : m_graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    }
    m_graphs.clear();
}

void CodecBase::quiesce()
{
    {
        std::unique_lock<std::mutex> lk(m_mu);
        m_cv_done.wait(lk, [&] { return m_done && !m_pending; });
    }
    if (m_cs) (void)hipStreamSynchronize(m_cs);
    if (m_io_stream) (void)hipStreamSynchronize(m_io_stream);
}

void CodecBase::bind_stage_arg(int key, const void* arg)
{
    GraphSlot& slot = m_graphs[key];
    if (slot.exec && slot.arg != arg) {
        (void)hipGraphExecDestroy(slot.exec);
        slot.exec = nullptr;
    }
    slot.arg = arg;
}

namespace {
// How results are handed to a caller that lives on the legacy NULL stream (torch's default stream;
// the reference harness installs a non-default one, test_video.py:423-425, plain torch code does not). hipStreamWaitEvent(NULL stream, event) is effectively host-blocking
// on ROCm 7.2: measured on MI355X (tools/ramp_probe.py, LD 1080p) 167 pictures/s with it against
// 277 with a torch side stream as the user stream - the host cannot run ahead of the GPU any more.
// Instead a BLOCKING stream of ours carries the wait: the legacy null stream orders every later
// operation of its own behind all blocking streams, so a consumer on it sees finished results
// without any call on the null stream itself (266 pictures/s; a clone() queued on the null stream
// right behind decompress, no host sync, saw the finished picture in 8 of 8 calls, and an
// unfinished one in 8 of 8 with no join at all). DCVC_NULL_STREAM_JOIN=event restores the plain wait.
bool join_by_blocking_stream()
{
    static const bool on = [] {
        const char* e = getenv("DCVC_NULL_STREAM_JOIN");
        return e == nullptr || std::string(e) != "event";
    }();
    return on;
}
}  // namespace

hipStream_t CodecBase::enter(hipStream_t user)
{
    hip_check(hipEventRecord(m_ev_in, user), "hipEventRecord(in)");
    hip_check(hipStreamWaitEvent(m_cs, m_ev_in, 0), "hipStreamWaitEvent(in)");
    return m_cs;
}

void CodecBase::leave(hipStream_t user)
{
    hip_check(hipEventRecord(m_ev_out, m_cs), "hipEventRecord(out)");
    if (user == nullptr && join_by_blocking_stream()) {
        if (m_join == nullptr) hip_check(hipStreamCreateWithFlags(&m_join, hipStreamDefault), "hipStreamCreate(join)");
        hip_check(hipStreamWaitEvent(m_join, m_ev_out, 0), "hipStreamWaitEvent(join)");
        return;
    }
    hip_check(hipStreamWaitEvent(user, m_ev_out, 0), "hipStreamWaitEvent(out)");
}

void CodecBase::submit(hipStream_t st, std::function<void()> job)
{
    hip_check(hipEventRecord(m_ev_job, st), "hipEventRecord(job)");
    hip_check(hipStreamWaitEvent(m_io_stream, m_ev_job, 0), "hipStreamWaitEvent(job)");
    {
        // a call that threw between submit() and wait_job() leaves its job running: never hand the
        // worker a second job (or reset its result) before the first has finished
        std::unique_lock<std::mutex> lk(m_mu);
        m_cv_done.wait(lk, [&] { return m_done && !m_pending; });
        m_job = std::move(job);
        m_pending = true;
        m_done = false;
        m_worker_error.clear();
    }
    m_cv_work.notify_one();
}

void CodecBase::wait_job()
{
    std::unique_lock<std::mutex> lk(m_mu);
    m_cv_done.wait(lk, [&] { return m_done; });
    if (!m_worker_error.empty()) throw std::runtime_error("entropy worker: " + m_worker_error);
}

void CodecBase::worker_loop()
{
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(m_mu);
            m_cv_work.wait(lk, [&] { return m_pending || m_stop; });
            if (m_stop) return;
            m_pending = false;
            job = std::move(m_job);
        }
        std::string err;
        static const bool trace = getenv("DCVC_TIMING") != nullptr;      // tuning aid: job time to stderr
        const auto t0 = std::chrono::steady_clock::now();
        try {
            job();
        } catch (const std::exception& e) {
            err = e.what();
        }
        if (trace) {
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "[dcvc] entropy job %.0f us, %zu bytes\n", us, m_enc.stream().size());
        }
        {
            std::lock_guard<std::mutex> lk(m_mu);
            m_worker_error = err;
            m_done = true;
        }
        m_cv_done.notify_all();
    }
}

void CodecBase::load_cdf_tables(const ParamStore& ps)
{
    auto cdf = [&](const char* name_cdf, const char* name_len, int index) {
        const HostTensor& c = ps.at(name_cdf);
        const HostTensor& l = ps.at(name_len);
        const int num = static_cast<int>(l.numel());
        const int stride = static_cast<int>(c.numel() / num);
        m_enc.set_cdf(c.i.data(), num, stride, l.i.data(), index);
        m_dec.set_cdf(c.i.data(), num, stride, l.i.data(), index);
    };
    cdf("bit_estimator_z.quantized_cdf", "bit_estimator_z.cdf_length", 0);
    cdf("gaussian_encoder.quantized_cdf", "gaussian_encoder.cdf_length", 1);
}

const half_t* CodecBase::upload_qp_table(const ParamStore& ps, DeviceArena& mem, const char* name, int ch)
{
    const HostTensor& t = ps.at(name);
    if (t.shape.size() != 2 || t.shape[0] != kQpNum || t.shape[1] != ch) {
        throw std::invalid_argument(std::string("unexpected shape for ") + name);
    }
    return mem.upload(t.h);
}

namespace {
struct QpRowsArg {
    half_t* dst[4];
    const half_t* src[4];
    int ch[4];
};

// block b copies row b: at most 4 rows of at most a few hundred channels - one tiny launch
__global__ void copy_qp_rows_kernel(const QpRowsArg a)
{
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < a.ch[r]; c += blockDim.x) a.dst[r][c] = a.src[r][c];
}
}  // namespace

void CodecBase::copy_qp_rows(std::initializer_list<QpRow> rows, int qp, hipStream_t st)
{
    if (qp < 0 || qp >= kQpNum) throw std::invalid_argument("qp out of range [0, 63]");
    if (rows.size() == 0 || rows.size() > 4) throw std::invalid_argument("copy_qp_rows: 1..4 rows");
    QpRowsArg a{};
    int n = 0;
    for (const QpRow& r : rows) {
        a.dst[n] = r.dst;
        a.src[n] = r.table + static_cast<size_t>(qp) * r.ch;
        a.ch[n] = r.ch;
        ++n;
    }
    hipLaunchKernelGGL(copy_qp_rows_kernel, dim3(n), dim3(256), 0, st, a);
    hip_check(hipGetLastError(), "copy_qp_rows");
}

}  // namespace dcvc
