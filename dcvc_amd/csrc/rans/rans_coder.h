// Host-side range-ANS entropy coder for the DCVC-UF bitstream (product code).
//
// Bitstream-compatible with the reference coder
//   /root/reference/src/cpp/py_rans/rans.cpp:13-19   (16-bit probabilities, 32-bit state,
//                                                     lower bound 2^23, byte renormalisation,
//                                                     2-bit bypass groups for escaped values)
//   /root/reference/src/cpp/py_rans/py_rans.cpp:156-249, 412-492  (sub-stream container)
// but organised differently: the coder is a plain object with flat tables, segments are
// queued and all sub-streams are coded by a small worker pool at flush()/wait() time
// (the reference keeps one thread + queue per sub-stream object).
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <cstdlib>
#include <new>
#include <thread>
#include <vector>

namespace dcvc {

constexpr int kMaxEcParallel = 8;       // reference: py_rans.h:15
constexpr int kRansProbBits = 16;       // rans.cpp:13
constexpr uint32_t kRansLow = 1u << 23; // rans.cpp:15
constexpr int kBypassBits = 2;          // rans.cpp:18

// One family of CDFs (table 0 = factorised z prior, table 1 = Gaussian y prior).
struct CdfTable {
    int num = 0;                       // number of CDFs
    int stride = 0;                    // entries per CDF row
    std::vector<uint16_t> start;       // [num*stride] cumulative start of value v
    std::vector<uint16_t> freq;        // [num*stride] frequency of value v
    std::vector<uint32_t> rcp;         // [num*stride] Alverson reciprocal of freq
    std::vector<uint8_t> rcp_shift;    // [num*stride]
    std::vector<uint32_t> cdf;         // [num*stride] raw cumulative values (decoder)
    std::vector<int8_t> max_value;     // [num] escape value = cdf_length - 2
    // decoder search start: first[i*256 + (cum >> 8)] = the value whose interval holds (cum & ~255), so
    // the scan for cum starts there instead of at 0 (built for families of at most kLutRows rows: the
    // 128 Gaussian tables = 32 KB, L1-resident; the per-channel z tables code ~1 % of the symbols)
    static constexpr int kLutRows = 256;
    std::vector<uint8_t> first;
    // AVX-512 search rows (hosts with AVX-512BW): edge[i * kEdgeRow + j - 1] = cdf[i][j] - 1, 0xFFFF beyond the row
    static constexpr int kEdgeRow = 32;
    template <class T> struct Aligned64 {             // rows must not straddle cache lines; copies of a table stay aligned
        using value_type = T;
        Aligned64() = default;
        template <class U> Aligned64(const Aligned64<U>&) {}
        T* allocate(size_t n) {
            void* p = std::aligned_alloc(64, (n * sizeof(T) + 63) / 64 * 64);
            if (!p) throw std::bad_alloc();
            return static_cast<T*>(p);
        }
        void deallocate(T* p, size_t) { std::free(p); }
        template <class U> bool operator==(const Aligned64<U>&) const { return true; }
        template <class U> bool operator!=(const Aligned64<U>&) const { return false; }
    };
    std::vector<uint16_t, Aligned64<uint16_t>> edge;      // empty = portable search
    void load(const int32_t* cdfs, int num_cdf, int row_stride, const int32_t* cdf_sizes);
};

// Fixed-size pool: run(n, fn) executes fn(0..n-1) on persistent threads and returns when
// all are done.
class WorkerPool {
public:
    explicit WorkerPool(int threads);
    ~WorkerPool();
    void run(int n, const std::function<void(int)>& fn);

private:
    void loop(int tid);
    std::vector<std::thread> m_threads;
    std::mutex m_mu;
    std::condition_variable m_cv_work, m_cv_done;
    const std::function<void(int)>* m_fn = nullptr;
    int m_n = 0, m_next = 0, m_pending = 0;
    uint64_t m_epoch = 0;
    bool m_stop = false;
    // lock-free mirrors of m_epoch / m_pending for the bounded spin in front of the condition-variable waits:
    // a picture is decoded in five short run() calls a few hundred microseconds apart, and a sleeping worker
    // costs 50-90 us of wake-up latency per call (measured) - more than the coding work of a small step
    std::atomic<uint64_t> m_epoch_hint{0};
    std::atomic<int> m_pending_hint{0};
    int m_spin_us = 1000;              // DCVC_RANS_SPIN_US; 0 = always sleep
    int m_spin_workers = 4;            // DCVC_RANS_SPIN_WORKERS: workers that spin that long (the rest: <= 100 us)
    std::exception_ptr m_error, m_local_error;      // first failure of a run(), rethrown by run()
    std::mutex m_err_mu;
};

class RansEncoder {
public:
    RansEncoder();
    void set_cdf(const int32_t* cdfs, int num_cdf, int stride, const int32_t* cdf_sizes, int index);
    void set_parallel(int n);
    int parallel() const { return m_n; }
    void reset();
    // Segments are referenced, not copied: the caller keeps them alive until flush() returns.
    void push_y(const int16_t* symbols, int count);                       // (sym << 8) + cdf index
    void push_z(const int8_t* symbols, int count, int cdf_offset, int ch); // cdf = i % ch + offset
    void flush();                             // codes everything, builds the container
    const std::vector<uint8_t>& stream() const { return m_out; }

private:
    struct Segment {
        const int16_t* y = nullptr;
        const int8_t* z = nullptr;
        int count = 0, cdf_offset = 0, ch = 1;
    };
    void encode_substream(int i);
    CdfTable m_tab[2];
    std::vector<Segment> m_segs;
    std::vector<std::vector<uint8_t>> m_buf;   // per sub-stream scratch (filled from the back)
    std::vector<size_t> m_begin;               // first valid byte in m_buf[i]
    std::vector<uint8_t> m_out;
    int m_n = 1;
    WorkerPool m_pool;
};

class RansDecoder {
public:
    RansDecoder();
    void set_cdf(const int32_t* cdfs, int num_cdf, int stride, const int32_t* cdf_sizes, int index);
    void set_parallel(int n);
    void set_stream(const uint8_t* data, size_t size);   // copies what it needs
    // Decode `count` symbols into out[0..count); blocking.
    void decode_y(const uint8_t* indexes, int count, int8_t* out);
    void decode_z(int count, int cdf_offset, int ch, int8_t* out);

private:
    struct Sub {
        std::vector<uint8_t> bytes;
        const uint8_t* ptr = nullptr;
        const uint8_t* end = nullptr;
        uint32_t state = 0;
    };
    CdfTable m_tab[2];
    Sub m_sub[kMaxEcParallel];
    int m_n = 1;
    WorkerPool m_pool;
};

// ryg-style float pmf -> 16-bit cumulative table (py_rans.cpp:35-94).
std::vector<uint32_t> pmf_to_quantized_cdf(const float* pmf, int n);

}  // namespace dcvc
