// Host rANS coder, see rans_coder.h. Bitstream semantics follow
// /root/reference/src/cpp/py_rans/rans.cpp:31-181 (state update, renormalisation, bypass
// groups, value mapping) and py_rans.cpp:104-249,412-492 (splitting + container).
#include "rans_coder.h"

#include <chrono>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstdlib>

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <stdexcept>

namespace dcvc {

namespace {

constexpr uint32_t kProbMask = (1u << kRansProbBits) - 1;
constexpr int kEncRenormShift = 23 - kRansProbBits + 8;        // rans.cpp:16
constexpr uint32_t kBypassMax = (1u << kBypassBits) - 1;       // rans.cpp:19
constexpr uint32_t kBypassFreqLimit = (1u << (kRansProbBits - kBypassBits)) << kEncRenormShift;

// ---------------------------------------------------------------- encoder primitives
struct EncState {
    uint32_t r;
    uint8_t* p;   // next free byte is p[-1]; bytes are produced back to front
};

inline void enc_put(EncState& s, uint32_t start, uint32_t freq, uint32_t rcp, uint32_t rcp_shift)
{
    const uint32_t r_max = freq << kEncRenormShift;
    uint32_t r = s.r;
    while (r >= r_max) {
        *--s.p = static_cast<uint8_t>(r);
        r >>= 8;
    }
    // q = r / freq through an exact Alverson reciprocal (valid for r < 2^31, which the
    // renormalisation above guarantees); freq == 1 uses rcp = 2^32-1, giving q = r - 1 and the
    // correction below restores r << 16.
    uint32_t q;
    if (freq == 1) {
        s.r = (r << kRansProbBits) + start;
        return;
    }
    q = static_cast<uint32_t>((static_cast<uint64_t>(r) * rcp) >> 32) >> rcp_shift;
    s.r = (q << kRansProbBits) + (r - q * freq) + start;
}

inline void enc_put_bits(EncState& s, uint32_t val)
{
    uint32_t r = s.r;
    while (r >= kBypassFreqLimit) {
        *--s.p = static_cast<uint8_t>(r);
        r >>= 8;
    }
    s.r = (r << kBypassBits) | val;
}

inline void enc_symbol(EncState& s, int32_t sym, const CdfTable& t, int cdf_idx)
{
    const int32_t max_value = t.max_value[cdf_idx];
    int32_t value = (sym < 0 ? -sym : sym) * 2 - (sym > 0);
    if (value >= max_value) {
        const uint32_t raw = static_cast<uint32_t>(value - max_value);
        value = max_value;
        // rANS is last-in-first-out: emit the raw groups (high group first), then the unary
        // group count, so the decoder meets count first and the low group of raw next.
        int n_groups = 0;
        while ((raw >> (n_groups * kBypassBits)) != 0) {
            ++n_groups;
        }
        for (int j = n_groups - 1; j >= 0; --j) {
            enc_put_bits(s, (raw >> (j * kBypassBits)) & kBypassMax);
        }
        const int n_full = n_groups / static_cast<int>(kBypassMax);
        enc_put_bits(s, static_cast<uint32_t>(n_groups) - n_full * kBypassMax);
        for (int j = 0; j < n_full; ++j) {
            enc_put_bits(s, kBypassMax);
        }
    }
    const size_t e = static_cast<size_t>(cdf_idx) * t.stride + value;
    enc_put(s, t.start[e], t.freq[e], t.rcp[e], t.rcp_shift[e]);
}

// ---------------------------------------------------------------- decoder primitives
struct DecState {
    uint32_t r;
    const uint8_t* p;
    const uint8_t* end;
};

inline uint32_t dec_byte(DecState& s)
{
    // A well-formed stream never runs dry; zeros keep a damaged one from reading outside.
    return s.p != s.end ? *s.p++ : 0u;
}

inline uint32_t dec_bits(DecState& s)
{
    // one 2-bit bypass group; the refill (one byte every fourth group of a run) is a select, not a branch: the
    // escape-coded symbols of the low-qp streams take 5-8 groups each, and a quarter of those branches mispredict
    const uint32_t val = s.r & kBypassMax;
    const uint32_t r = s.r >> kBypassBits;
    const bool more = s.p != s.end;                       // a well-formed stream never runs dry
    const uint32_t byte = more ? *s.p : 0u;
    const uint32_t need = static_cast<uint32_t>(r < kRansLow);
    s.r = need ? ((r << 8) | byte) : r;
    s.p += need & static_cast<uint32_t>(more);
    return val;
}

// Everything of a symbol behind the CDF search: state update, byte renormalisation, escape groups.
inline int8_t dec_finish(DecState& s, const uint32_t* cdf, int v, uint32_t cum, int32_t max_value)
{
    const uint32_t start = cdf[v];
    const uint32_t freq = cdf[v + 1] - start;
    uint32_t r = freq * (s.r >> kRansProbBits) + cum - start;
    // renormalisation without a data-dependent branch: r >= 2^7 here (freq >= 1, state >= 2^23), so 0, 1
    // or 2 bytes bring it back above 2^23; whether a byte is needed is close to a coin flip per symbol
    if (s.end - s.p >= 2) {
        const uint32_t n = static_cast<uint32_t>(r < kRansLow) + static_cast<uint32_t>(r < (kRansLow >> 8));
        const uint32_t two = (static_cast<uint32_t>(s.p[0]) << 8) | s.p[1];
        r = (r << (8 * n)) | (two >> (16 - 8 * n));
        s.p += n;
    } else {
        while (r < kRansLow) {
            r = (r << 8) | dec_byte(s);
        }
    }
    s.r = r;
    int32_t value = v;
    if (__builtin_expect(value == max_value, 0)) {
        uint32_t g = dec_bits(s);
        uint32_t n_groups = g;
        while (g == kBypassMax) {
            g = dec_bits(s);
            n_groups += g;
            if (n_groups > 16) {
                // an int8 payload needs at most 4 groups; a damaged stream must not shift by >= 32
                throw std::runtime_error("rANS decoder: corrupt escape code");
            }
        }
        uint32_t raw = 0;
        for (uint32_t j = 0; j < n_groups; ++j) {
            raw |= dec_bits(s) << (j * kBypassBits);
        }
        value = static_cast<int32_t>(raw) + max_value;
    }
    const int32_t mag = (value + 1) >> 1;
    return static_cast<int8_t>((value & 1) ? mag : -mag);
}

// Portable search: start at the value whose interval holds (cum & ~255), scan upwards.
inline int8_t dec_symbol(DecState& s, const CdfTable& t, int cdf_idx)
{
    const uint32_t* cdf = t.cdf.data() + static_cast<size_t>(cdf_idx) * t.stride;
    const uint32_t cum = s.r & kProbMask;
    int v = t.first.empty() ? 0 : t.first[static_cast<size_t>(cdf_idx) * 256 + (cum >> 8)];
    while (cdf[v + 1] <= cum) {
        ++v;
    }
    return dec_finish(s, cdf, v, cum, t.max_value[cdf_idx]);
}

#if defined(__x86_64__)
// AVX-512 search (round 4): a CDF row of the reference's tables has at most 19 entries, so the whole
// row sits in ONE 512-bit register as 32 x u16 (`edge`: cdf[j] - 1 for j = 1 .. size - 1, 0xFFFF beyond; the terminal
// 2^16 becomes 0xFFFF, which no 16-bit cum exceeds): value = number of j >= 1 with cdf[j] <= cum = popcount of ONE
// unsigned compare - no scan loop and, above all, no data-dependent loop exit to mispredict (the exit of the scan above
// is close to a coin flip per symbol on spread distributions). The row's address depends on the index stream only, so
// its load runs ahead of the serial state chain. Same arithmetic, same bytes: tests/test_rans.py decodes the golden streams
// in child processes with DCVC_RANS_AVX512=0 and =1 (test_both_cdf_search_paths).
#define DCVC_RANS_AVX512 1
__attribute__((target("avx512f,avx512bw,popcnt"))) inline int search_avx512(const uint16_t* edge_row, uint32_t cum)
{
    const __m512i row = _mm512_load_si512(reinterpret_cast<const void*>(edge_row));
    const __mmask32 le = _mm512_cmplt_epu16_mask(row, _mm512_set1_epi16(static_cast<short>(cum)));
    return static_cast<int>(_mm_popcnt_u32(static_cast<unsigned>(le)));
}

__attribute__((target("avx512f,avx512bw,popcnt")))
void decode_y_avx512(DecState& st, const CdfTable& t, const uint8_t* indexes, int b, int len, int8_t* out)
{
    const uint32_t* const cdf0 = t.cdf.data();
    const uint16_t* const edge = t.edge.data();
    const size_t stride = static_cast<size_t>(t.stride);
    for (int k = b; k < b + len; ++k) {
        const int idx = indexes[k];
        const uint32_t cum = st.r & kProbMask;
        const int v = search_avx512(edge + static_cast<size_t>(idx) * CdfTable::kEdgeRow, cum);
        out[k] = dec_finish(st, cdf0 + idx * stride, v, cum, t.max_value[idx]);
    }
}

__attribute__((target("avx512f,avx512bw,popcnt")))
void decode_z_avx512(DecState& st, const CdfTable& t, int cdf_offset, int ch, int b, int len, int8_t* out)
{
    const uint32_t* const cdf0 = t.cdf.data();
    const uint16_t* const edge = t.edge.data();
    const size_t stride = static_cast<size_t>(t.stride);
    for (int k = b; k < b + len; ++k) {
        const int idx = (k % ch) + cdf_offset;
        const uint32_t cum = st.r & kProbMask;
        const int v = search_avx512(edge + static_cast<size_t>(idx) * CdfTable::kEdgeRow, cum);
        out[k] = dec_finish(st, cdf0 + idx * stride, v, cum, t.max_value[idx]);
    }
}

bool cpu_has_avx512bw()
{
    static const bool have = [] {
        if (const char* e = getenv("DCVC_RANS_AVX512")) {
            if (atoi(e) == 0) return false;             // A/B switch and the portable path's test hook
        }
        __builtin_cpu_init();
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("popcnt");
    }();
    return have;
}
#else
bool cpu_has_avx512bw() { return false; }
#endif

inline void slice_of(int count, int n, int i, int& begin, int& len)
{
    const int size0 = count / n;
    begin = size0 * i;
    len = (i == n - 1) ? count - size0 * (n - 1) : size0;
}

// The reference lets a pair of streams share their trailing bytes when both end in zeros
// (py_rans.cpp:13-33).
int shared_tail_bytes(const uint8_t* a, size_t na, const uint8_t* b, size_t nb)
{
    int same = 0;
    const size_t check = std::min<size_t>({ na, nb, 8 });
    for (size_t i = 0; i < check; ++i) {
        if (a[na - 1 - i] != 0 || b[nb - 1 - i] != 0) {
            break;
        }
        ++same;
    }
    if (same == 0 && na > 0 && nb > 0 && a[na - 1] == b[nb - 1]) {
        same = 1;
    }
    return same;
}

void store_le32(uint8_t* p, int32_t v)
{
    const uint32_t u = static_cast<uint32_t>(v);
    p[0] = static_cast<uint8_t>(u);
    p[1] = static_cast<uint8_t>(u >> 8);
    p[2] = static_cast<uint8_t>(u >> 16);
    p[3] = static_cast<uint8_t>(u >> 24);
}

int32_t load_le32(const uint8_t* p)
{
    return static_cast<int32_t>(p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24));
}

}  // namespace

// ------------------------------------------------------------------------ CdfTable
void CdfTable::load(const int32_t* cdfs, int num_cdf, int row_stride, const int32_t* cdf_sizes)
{
    num = num_cdf;
    stride = row_stride;
    const size_t total = static_cast<size_t>(num) * stride;
    start.assign(total, 0);
    freq.assign(total, 0);
    rcp.assign(total, 0);
    rcp_shift.assign(total, 0);
    cdf.assign(total + 1, 0);
    max_value.assign(num, 0);
    for (int i = 0; i < num; ++i) {
        max_value[i] = static_cast<int8_t>(cdf_sizes[i] - 2);
        const int32_t* row = cdfs + static_cast<size_t>(i) * stride;
        for (int j = 0; j < stride; ++j) {
            cdf[static_cast<size_t>(i) * stride + j] = static_cast<uint32_t>(row[j]);
        }
        for (int j = 0; j + 1 < stride; ++j) {
            const size_t e = static_cast<size_t>(i) * stride + j;
            const uint32_t f = static_cast<uint16_t>(row[j + 1] - row[j]);
            start[e] = static_cast<uint16_t>(row[j]);
            freq[e] = static_cast<uint16_t>(f);
            if (f >= 2) {
                uint32_t shift = 0;
                while (f > (1u << shift)) {
                    ++shift;
                }
                rcp[e] = static_cast<uint32_t>(((1ull << (shift + 31)) + f - 1) / f);
                rcp_shift[e] = static_cast<uint8_t>(shift - 1);
            }
        }
    }
    // AVX-512 search rows (see search_avx512): 32 x u16 per CDF, 64-byte aligned
    edge.clear();
    // The vector search counts j >= 1 with cdf[j] - 1 < cum (16-bit): it equals the scan only for rows that rise
    // monotonically from cdf[0] = 0 with cdf[j] >= 1 for j >= 1 - a zero there (a leading zero-frequency symbol) would wrap
    // to 0xFFFF and never be counted, while the scan counts it (advisor, round 4). The reference's tables are of that form
    // (pmf_to_quantized_cdf gives every symbol a frequency >= 1); a table that is not keeps the portable search.
    bool vector_ok = stride <= kEdgeRow + 1 && kRansProbBits == 16 && cpu_has_avx512bw();
    for (int i = 0; vector_ok && i < num; ++i) {
        const uint32_t* row = cdf.data() + static_cast<size_t>(i) * stride;
        const int size = cdf_sizes[i];
        if (size < 2 || size > stride || row[0] != 0) vector_ok = false;
        for (int j = 1; vector_ok && j < size; ++j) {
            if (row[j] == 0 || row[j] < row[j - 1] || row[j] > (1u << kRansProbBits)) vector_ok = false;
        }
        // ... and ends at the full range: a row that does not is malformed, and the two searches would decode it differently
        if (vector_ok && row[size - 1] != (1u << kRansProbBits)) vector_ok = false;
    }
    // one non-conforming row keeps the WHOLE table on the portable search: said once, so that the slower decode path is visible
    if (!vector_ok && stride <= kEdgeRow + 1 && kRansProbBits == 16 && cpu_has_avx512bw() && getenv("DCVC_RANS_AVX512") == nullptr) {
        static bool said = false;
        if (!said) {
            said = true;
            fprintf(stderr, "[dcvc] rANS decoder: a CDF table (%d rows) is not of the form the vector search needs "
                            "(0 = cdf[0] < cdf[1] <= ... <= cdf[size - 1] = 65536): portable search for it\n", num);
        }
    }
    if (vector_ok) {
        edge.assign(static_cast<size_t>(num) * kEdgeRow, 0xFFFFu);
        for (int i = 0; i < num; ++i) {
            const uint32_t* row = cdf.data() + static_cast<size_t>(i) * stride;
            const int size = cdf_sizes[i];                  // entries of the row: 0, ..., 2^16
            for (int j = 1; j < size && j <= kEdgeRow; ++j) {
                edge[static_cast<size_t>(i) * kEdgeRow + (j - 1)] = static_cast<uint16_t>(row[j] - 1u);
            }
        }
    }
    first.clear();
    if (num <= kLutRows && kRansProbBits == 16) {
        first.assign(static_cast<size_t>(num) * 256, 0);
        for (int i = 0; i < num; ++i) {
            const uint32_t* row = cdf.data() + static_cast<size_t>(i) * stride;
            const int last = cdf_sizes[i] - 2;              // the scan never has to pass the escape value
            int v = 0;
            for (uint32_t b = 0; b < 256; ++b) {
                while (v < last && row[v + 1] <= (b << 8)) ++v;
                first[static_cast<size_t>(i) * 256 + b] = static_cast<uint8_t>(v);
            }
        }
    }
}

// ------------------------------------------------------------------------ WorkerPool
namespace {
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}
}  // namespace

WorkerPool::WorkerPool(int threads)
{
    // Wake-up latency of a sleeping worker (50-100 us) is the size of a whole entropy call, so workers stay hot for a
    // while after a run. Bounded on purpose: only the first DCVC_RANS_SPIN_WORKERS (default 4)
    // workers spin for the full DCVC_RANS_SPIN_US (default 1000 us = the distance between the 4-5 entropy calls of a
    // picture); the others give up after 100 us, so a node with one codec per GPU burns at most 4 cores per rank.
    if (const char* e = getenv("DCVC_RANS_SPIN_US")) m_spin_us = atoi(e);
    if (const char* e = getenv("DCVC_RANS_SPIN_WORKERS")) m_spin_workers = atoi(e);
    for (int i = 0; i < threads; ++i) {
        m_threads.emplace_back(&WorkerPool::loop, this, i);
    }
}

WorkerPool::~WorkerPool()
{
    {
        std::lock_guard<std::mutex> lk(m_mu);
        m_stop = true;
    }
    m_cv_work.notify_all();
    for (auto& t : m_threads) {
        t.join();
    }
}

void WorkerPool::loop(int index)
{
    const int spin_us = index < m_spin_workers ? m_spin_us : std::min(m_spin_us, 100);
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(m_mu);
    for (;;) {
        m_cv_work.wait(lk, [&] { return m_stop || (m_epoch != seen && m_next < m_n); });
        if (m_stop) {
            return;
        }
        while (m_next < m_n) {
            const int item = m_next++;
            const auto* fn = m_fn;
            lk.unlock();
            std::exception_ptr err;
            try {
                (*fn)(item);
            } catch (...) {
                err = std::current_exception();
            }
            lk.lock();
            if (err && !m_error) {
                m_error = err;
            }
            m_pending_hint.store(--m_pending, std::memory_order_release);
            if (m_pending == 0) {
                m_cv_done.notify_all();
            }
        }
        seen = m_epoch;
        if (spin_us > 0 && !m_stop) {
            // stay hot for a while: the next run() of the same picture is usually close
            lk.unlock();
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
            while (m_epoch_hint.load(std::memory_order_acquire) == seen) {
                for (int i = 0; i < 64; ++i) cpu_relax();
                if (std::chrono::steady_clock::now() >= until) break;
            }
            lk.lock();
        }
    }
}

void WorkerPool::run(int n, const std::function<void(int)>& fn)
{
    if (n <= 0) {
        return;
    }
    if (n == 1 || m_threads.empty()) {
        for (int i = 0; i < n; ++i) {
            fn(i);
        }
        return;
    }
    std::unique_lock<std::mutex> lk(m_mu);
    m_fn = &fn;
    m_n = n;
    m_next = 1;          // item 0 runs on the calling thread
    m_pending = n;
    m_error = nullptr;
    ++m_epoch;
    m_pending_hint.store(n, std::memory_order_release);
    m_epoch_hint.store(m_epoch, std::memory_order_release);
    lk.unlock();
    m_cv_work.notify_all();
    // a failing item (bad_alloc, corrupt stream) must neither terminate a pool thread nor leave
    // run() before every item has finished with the caller's buffers: first error wins, rethrown below
    auto guarded = [&](int item) {
        try {
            fn(item);
        } catch (...) {
            std::lock_guard<std::mutex> g(m_err_mu);
            if (!m_local_error) m_local_error = std::current_exception();
        }
    };
    m_local_error = nullptr;
    guarded(0);
    lk.lock();
    m_pending_hint.store(--m_pending, std::memory_order_release);
    // help with whatever the workers have not picked up yet
    while (m_next < m_n) {
        const int item = m_next++;
        lk.unlock();
        guarded(item);
        lk.lock();
        m_pending_hint.store(--m_pending, std::memory_order_release);
    }
    if (m_pending != 0 && m_spin_us > 0) {
        lk.unlock();
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(m_spin_us);
        while (m_pending_hint.load(std::memory_order_acquire) > 0) {
            for (int i = 0; i < 64; ++i) cpu_relax();
            if (std::chrono::steady_clock::now() >= until) break;
        }
        lk.lock();
    }
    m_cv_done.wait(lk, [&] { return m_pending == 0; });
    m_fn = nullptr;
    m_n = 0;
    std::exception_ptr err = m_local_error ? m_local_error : m_error;
    m_error = nullptr;
    m_local_error = nullptr;
    lk.unlock();
    if (err) {
        std::rethrow_exception(err);
    }
}

// ------------------------------------------------------------------------ RansEncoder
RansEncoder::RansEncoder() : m_pool(kMaxEcParallel - 1)
{
    m_buf.resize(kMaxEcParallel);
    m_begin.assign(kMaxEcParallel, 0);
}

void RansEncoder::set_cdf(const int32_t* cdfs, int num_cdf, int stride, const int32_t* cdf_sizes,
                          int index)
{
    if (index < 0 || index > 1) {
        throw std::invalid_argument("rANS table index must be 0 (z) or 1 (y)");
    }
    m_tab[index].load(cdfs, num_cdf, stride, cdf_sizes);
}

void RansEncoder::set_parallel(int n)
{
    if (n < 1 || n > kMaxEcParallel) {
        throw std::invalid_argument("entropy coder parallelism must be in [1, 8]");
    }
    m_n = n;
}

void RansEncoder::reset()
{
    m_segs.clear();
    m_out.clear();
}

void RansEncoder::push_y(const int16_t* symbols, int count)
{
    Segment s;
    s.y = symbols;
    s.count = count;
    m_segs.push_back(s);
}

void RansEncoder::push_z(const int8_t* symbols, int count, int cdf_offset, int ch)
{
    Segment s;
    s.z = symbols;
    s.count = count;
    s.cdf_offset = cdf_offset;
    s.ch = ch;
    m_segs.push_back(s);
}

void RansEncoder::encode_substream(int i)
{
    size_t symbols = 0;
    for (const Segment& seg : m_segs) {
        int b, len;
        slice_of(seg.count, m_n, i, b, len);
        symbols += static_cast<size_t>(len);
    }
    // <= 30 bits per symbol (16 for the modelled value, <= 14 for an escaped int8), plus state.
    const size_t cap = symbols * 4 + 16;
    std::vector<uint8_t>& buf = m_buf[i];
    if (buf.size() < cap) {
        buf.resize(cap);
    }
    EncState st;
    st.r = kRansLow;
    st.p = buf.data() + buf.size();
    for (const Segment& seg : m_segs) {
        int b, len;
        slice_of(seg.count, m_n, i, b, len);
        if (seg.y != nullptr) {
            const CdfTable& t = m_tab[1];
            for (int k = b + len - 1; k >= b; --k) {
                const int16_t c = seg.y[k];
                enc_symbol(st, static_cast<int8_t>(c >> 8), t, c & 0xff);
            }
        } else {
            const CdfTable& t = m_tab[0];
            for (int k = b + len - 1; k >= b; --k) {
                enc_symbol(st, seg.z[k], t, (k % seg.ch) + seg.cdf_offset);
            }
        }
    }
    st.p -= 4;
    st.p[0] = static_cast<uint8_t>(st.r);
    st.p[1] = static_cast<uint8_t>(st.r >> 8);
    st.p[2] = static_cast<uint8_t>(st.r >> 16);
    st.p[3] = static_cast<uint8_t>(st.r >> 24);
    m_begin[i] = static_cast<size_t>(st.p - buf.data());
}

void RansEncoder::flush()
{
    const int n = m_n;
    m_pool.run(n, [this](int i) { encode_substream(i); });

    auto sub = [&](int i) { return m_buf[i].data() + m_begin[i]; };
    auto len = [&](int i) { return m_buf[i].size() - m_begin[i]; };

    if (n == 1) {
        m_out.assign(sub(0), sub(0) + len(0));
        return;
    }
    const int pairs = n / 2;
    const bool tail = (n % 2) != 0;
    std::vector<int> shared(pairs), group(pairs);
    for (int p = 0; p < pairs; ++p) {
        shared[p] = shared_tail_bytes(sub(2 * p), len(2 * p), sub(2 * p + 1), len(2 * p + 1));
        group[p] = static_cast<int>(len(2 * p) + len(2 * p + 1)) - shared[p];
    }
    const int n_offsets = pairs - 1 + (tail ? 1 : 0);
    size_t total = static_cast<size_t>(n_offsets) * 4;
    for (int p = 0; p < pairs; ++p) {
        total += static_cast<size_t>(group[p]);
    }
    if (tail) {
        total += len(n - 1);
    }
    m_out.resize(total);
    uint8_t* out = m_out.data();
    int cumulative = 0;
    for (int k = 0; k < n_offsets; ++k) {
        cumulative += group[k];
        store_le32(out + 4 * k, cumulative);
    }
    size_t pos = static_cast<size_t>(n_offsets) * 4;
    for (int p = 0; p < pairs; ++p) {
        const size_t na = len(2 * p), nb = len(2 * p + 1);
        std::memcpy(out + pos, sub(2 * p), na);
        std::reverse_copy(sub(2 * p + 1), sub(2 * p + 1) + (nb - shared[p]), out + pos + na);
        pos += static_cast<size_t>(group[p]);
    }
    if (tail) {
        std::memcpy(out + pos, sub(n - 1), len(n - 1));
    }
}

// ------------------------------------------------------------------------ RansDecoder
RansDecoder::RansDecoder() : m_pool(kMaxEcParallel - 1) {}

void RansDecoder::set_cdf(const int32_t* cdfs, int num_cdf, int stride, const int32_t* cdf_sizes,
                          int index)
{
    if (index < 0 || index > 1) {
        throw std::invalid_argument("rANS table index must be 0 (z) or 1 (y)");
    }
    m_tab[index].load(cdfs, num_cdf, stride, cdf_sizes);
}

void RansDecoder::set_parallel(int n)
{
    if (n < 1 || n > kMaxEcParallel) {
        throw std::invalid_argument("entropy coder parallelism must be in [1, 8]");
    }
    m_n = n;
}

void RansDecoder::set_stream(const uint8_t* data, size_t size)
{
    const int n = m_n;
    auto open = [&](int i, const uint8_t* p, size_t len, bool reversed) {
        Sub& s = m_sub[i];
        s.bytes.resize(len);
        if (reversed) {
            std::reverse_copy(p, p + len, s.bytes.data());
        } else if (len > 0) {
            std::memcpy(s.bytes.data(), p, len);
        }
        s.ptr = s.bytes.data();
        s.end = s.ptr + len;
        uint32_t r = 0;
        for (int k = 0; k < 4; ++k) {
            r |= (s.ptr != s.end ? static_cast<uint32_t>(*s.ptr++) : 0u) << (8 * k);
        }
        s.state = r;
    };
    if (n == 1) {
        open(0, data, size, false);
        return;
    }
    if (n == 2) {
        open(0, data, size, false);
        open(1, data, size, true);
        return;
    }
    const int pairs = n / 2;
    const bool tail = (n % 2) != 0;
    const int n_offsets = pairs - 1 + (tail ? 1 : 0);
    const size_t header = static_cast<size_t>(n_offsets) * 4;
    if (size < header) {
        throw std::runtime_error("rANS container shorter than its offset header");
    }
    std::vector<int64_t> bound(n_offsets);
    for (int k = 0; k < n_offsets; ++k) {
        bound[k] = load_le32(data + 4 * k);
    }
    const uint8_t* payload = data + header;
    const int64_t payload_size = static_cast<int64_t>(size - header);
    for (int p = 0; p < pairs; ++p) {
        const int64_t b = p == 0 ? 0 : bound[p - 1];
        const int64_t e = p < n_offsets ? bound[p] : payload_size;
        if (b < 0 || e < b || e > payload_size) {
            throw std::runtime_error("rANS container offsets are inconsistent");
        }
        open(2 * p, payload + b, static_cast<size_t>(e - b), false);
        open(2 * p + 1, payload + b, static_cast<size_t>(e - b), true);
    }
    if (tail) {
        const int64_t b = bound[n_offsets - 1];
        if (b < 0 || b > payload_size) {
            throw std::runtime_error("rANS container offsets are inconsistent");
        }
        open(n - 1, payload + b, static_cast<size_t>(payload_size - b), false);
    }
}

void RansDecoder::decode_y(const uint8_t* indexes, int count, int8_t* out)
{
    const int n = m_n;
    m_pool.run(n, [&](int i) {
        int b, len;
        slice_of(count, n, i, b, len);
        Sub& sub = m_sub[i];
        DecState st{ sub.state, sub.ptr, sub.end };
        const CdfTable& t = m_tab[1];
#ifdef DCVC_RANS_AVX512
        if (!t.edge.empty()) {
            decode_y_avx512(st, t, indexes, b, len, out);
        } else
#endif
        for (int k = b; k < b + len; ++k) {
            out[k] = dec_symbol(st, t, indexes[k]);
        }
        sub.state = st.r;
        sub.ptr = st.p;
    });
}

void RansDecoder::decode_z(int count, int cdf_offset, int ch, int8_t* out)
{
    const int n = m_n;
    m_pool.run(n, [&](int i) {
        int b, len;
        slice_of(count, n, i, b, len);
        Sub& sub = m_sub[i];
        DecState st{ sub.state, sub.ptr, sub.end };
        const CdfTable& t = m_tab[0];
#ifdef DCVC_RANS_AVX512
        if (!t.edge.empty()) {
            decode_z_avx512(st, t, cdf_offset, ch, b, len, out);
        } else
#endif
        for (int k = b; k < b + len; ++k) {
            out[k] = dec_symbol(st, t, (k % ch) + cdf_offset);
        }
        sub.state = st.r;
        sub.ptr = st.p;
    });
}

// ------------------------------------------------------------------------ pmf -> cdf
std::vector<uint32_t> pmf_to_quantized_cdf(const float* pmf, int n)
{
    constexpr uint32_t total_prob = 1u << kRansProbBits;
    std::vector<uint32_t> cdf(static_cast<size_t>(n) + 1, 0);
    for (int i = 0; i < n; ++i) {
        // float * int promoted to double by the + 0.5, as in the reference lambda
        cdf[i + 1] = static_cast<uint32_t>(pmf[i] * total_prob + 0.5);
    }
    uint32_t sum = 0;
    for (uint32_t v : cdf) {
        sum += v;
    }
    if (sum == 0) {
        throw std::invalid_argument("pmf sums to zero");
    }
    for (uint32_t& v : cdf) {
        v = static_cast<uint32_t>((static_cast<uint64_t>(total_prob) * v) / sum);
    }
    std::partial_sum(cdf.begin(), cdf.end(), cdf.begin());
    cdf.back() = total_prob;
    const int m = static_cast<int>(cdf.size());
    for (int i = 0; i + 1 < m; ++i) {
        if (cdf[i] + 1 > cdf[i + 1]) {
            // steal one count from the least frequent value that can spare it
            uint32_t best_freq = ~0u;
            int best = -1;
            for (int j = 0; j + 1 < m; ++j) {
                const uint32_t f = cdf[j + 1] - cdf[j];
                if (f >= 2 && f < best_freq) {
                    best_freq = f;
                    best = j;
                }
            }
            if (best < 0) {
                throw std::runtime_error("cannot give every value a non-zero frequency");
            }
            if (best < i) {
                for (int j = best + 1; j <= i; ++j) {
                    cdf[j] -= 1;
                }
            } else {
                for (int j = i + 1; j <= best; ++j) {
                    cdf[j] += 1;
                }
            }
        }
    }
    return cdf;
}

}  // namespace dcvc
