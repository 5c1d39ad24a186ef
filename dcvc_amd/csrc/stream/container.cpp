// See container.h / include/dcvc_amd_stream.h.
#include "stream/container.h"

#include "capi_common.h"
#include "dcvc_amd_stream.h"

#include <cstring>
#include <string>

namespace dcvc {

namespace stream {

void put_uint(std::vector<uint8_t>& out, uint32_t v)
{
    if (v >= (1u << 30)) throw StreamError(-1, "container: value does not fit a 30-bit varuint");
    if (v < (1u << 7)) {
        out.push_back(static_cast<uint8_t>(v));
    } else if (v < (1u << 14)) {
        out.push_back(static_cast<uint8_t>(0x80u | (v >> 8)));
        out.push_back(static_cast<uint8_t>(v));
    } else {
        out.push_back(static_cast<uint8_t>(0xC0u | (v >> 24)));
        out.push_back(static_cast<uint8_t>(v >> 16));
        out.push_back(static_cast<uint8_t>(v >> 8));
        out.push_back(static_cast<uint8_t>(v));
    }
}

void put_sps(std::vector<uint8_t>& out, int sps_id, int height, int width)
{
    if (sps_id < 0 || sps_id > 15 || height < 0 || width < 0) throw StreamError(-1, "container: SPS field out of range");
    out.push_back(static_cast<uint8_t>((kSps << 4) | sps_id));
    put_uint(out, static_cast<uint32_t>(height));
    put_uint(out, static_cast<uint32_t>(width));
}

void put_ip(std::vector<uint8_t>& out, bool is_i, int sps_id, int qp, int ec_part, bool reset,
            const uint8_t* payload, size_t n)
{
    if (sps_id < 0 || sps_id > 15 || qp < 0 || qp > 255 || ec_part < 0 || ec_part > 127) {
        throw StreamError(-1, "container: picture header field out of range");
    }
    if (n >= (1u << 30)) throw StreamError(-1, "container: payload too long for the length field");
    out.push_back(static_cast<uint8_t>(((is_i ? kIntra : kInter) << 4) | sps_id));
    out.push_back(static_cast<uint8_t>(qp));
    out.push_back(static_cast<uint8_t>((ec_part << 1) | (reset ? 1 : 0)));
    put_uint(out, static_cast<uint32_t>(n));
    out.insert(out.end(), payload, payload + n);
}

const uint8_t* Reader::take(size_t k)
{
    if (m_n - m_pos < k) throw StreamError(-2, "container: truncated stream");
    const uint8_t* q = m_p + m_pos;
    m_pos += k;
    return q;
}

uint32_t Reader::get_uint()
{
    const uint8_t first = *take(1);
    const int tag = first >> 6;
    if (tag < 2) return first;
    if (tag == 2) return (static_cast<uint32_t>(first & 0x3F) << 8) | *take(1);
    const uint8_t* r = take(3);
    return (static_cast<uint32_t>(first & 0x3F) << 24) | (static_cast<uint32_t>(r[0]) << 16) |
           (static_cast<uint32_t>(r[1]) << 8) | r[2];
}

void Reader::header(int& nal_type, int& sps_id)
{
    const uint8_t flag = *take(1);
    nal_type = flag >> 4;
    sps_id = flag & 0x0F;
    if (nal_type > kInter) throw StreamError(-3, "container: unknown unit type");
}

void Reader::sps_remaining(int& height, int& width)
{
    height = static_cast<int>(get_uint());
    width = static_cast<int>(get_uint());
}

void Reader::ip_remaining(int& qp, int& ec_part, bool& reset, const uint8_t*& payload, size_t& n)
{
    const uint8_t* h = take(2);
    qp = h[0];
    ec_part = h[1] >> 1;
    reset = (h[1] & 1) != 0;
    n = get_uint();
    payload = take(n);
}

int SpsTable::id_for(int height, int width, bool& is_new)
{
    int top = -1;
    for (const Sps& s : m_sets) {
        if (s.height == height && s.width == width) {
            is_new = false;
            return s.id;
        }
        if (s.id > top) top = s.id;
    }
    if (top + 1 > 15) throw StreamError(-1, "container: more than 16 picture sizes in one stream");
    m_sets.push_back(Sps{top + 1, height, width});
    is_new = true;
    return top + 1;
}

void SpsTable::add(int id, int height, int width)
{
    for (Sps& s : m_sets) {
        if (s.id == id) {
            s.height = height;
            s.width = width;
            return;
        }
    }
    m_sets.push_back(Sps{id, height, width});
}

const SpsTable::Sps* SpsTable::find(int id) const
{
    for (const Sps& s : m_sets) {
        if (s.id == id) return &s;
    }
    return nullptr;
}

}  // namespace stream
}  // namespace dcvc

// ------------------------------------------------------------------------------------ C ABI
namespace {

template <typename F>
int64_t guarded_stream(F&& fn)
{
    try {
        return fn();
    } catch (const dcvc::stream::StreamError& e) {
        dcvc::last_error() = e.what();
        return e.code;
    } catch (const std::exception& e) {
        dcvc::last_error() = e.what();
        return -3;
    }
}

int64_t emit(const std::vector<uint8_t>& v, uint8_t* dst, size_t cap)
{
    if (dst == nullptr) return static_cast<int64_t>(v.size());
    if (v.size() > cap) throw dcvc::stream::StreamError(-1, "container: destination too small");
    std::memcpy(dst, v.data(), v.size());
    return static_cast<int64_t>(v.size());
}

}  // namespace

extern "C" {

int dcvc_stream_write_uint(uint8_t* dst, size_t cap, uint32_t value)
{
    return static_cast<int>(guarded_stream([&] {
        std::vector<uint8_t> v;
        dcvc::stream::put_uint(v, value);
        return emit(v, dst, cap);
    }));
}

int dcvc_stream_read_uint(const uint8_t* src, size_t n, uint32_t* value)
{
    return static_cast<int>(guarded_stream([&] {
        dcvc::stream::Reader r(src, n);
        *value = r.get_uint();
        return static_cast<int64_t>(r.position());
    }));
}

int dcvc_stream_write_sps(uint8_t* dst, size_t cap, int sps_id, int height, int width)
{
    return static_cast<int>(guarded_stream([&] {
        std::vector<uint8_t> v;
        dcvc::stream::put_sps(v, sps_id, height, width);
        return emit(v, dst, cap);
    }));
}

int64_t dcvc_stream_write_ip(uint8_t* dst, size_t cap, int is_i_frame, int sps_id, int qp, int ec_part,
                             int reset_feature_memory, const uint8_t* payload, size_t payload_bytes)
{
    return guarded_stream([&] {
        std::vector<uint8_t> v;
        v.reserve(payload_bytes + 8);
        dcvc::stream::put_ip(v, is_i_frame != 0, sps_id, qp, ec_part, reset_feature_memory != 0, payload, payload_bytes);
        return emit(v, dst, cap);
    });
}

int dcvc_stream_read_header(const uint8_t* src, size_t n, int* nal_type, int* sps_id)
{
    return static_cast<int>(guarded_stream([&] {
        dcvc::stream::Reader r(src, n);
        r.header(*nal_type, *sps_id);
        return static_cast<int64_t>(r.position());
    }));
}

int dcvc_stream_read_sps_remaining(const uint8_t* src, size_t n, int* height, int* width)
{
    return static_cast<int>(guarded_stream([&] {
        dcvc::stream::Reader r(src, n);
        r.sps_remaining(*height, *width);
        return static_cast<int64_t>(r.position());
    }));
}

int64_t dcvc_stream_read_ip_remaining(const uint8_t* src, size_t n, int* qp, int* ec_part,
                                      int* reset_feature_memory, const uint8_t** payload, size_t* payload_bytes)
{
    return guarded_stream([&] {
        dcvc::stream::Reader r(src, n);
        bool reset = false;
        r.ip_remaining(*qp, *ec_part, reset, *payload, *payload_bytes);
        *reset_feature_memory = reset ? 1 : 0;
        return static_cast<int64_t>(r.position());
    });
}

}  // extern "C"
