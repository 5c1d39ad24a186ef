// Bit-stream container of a coded sequence (host code). Format and reference: include/dcvc_amd_stream.h
// (stream_helper.py:37-154). C++ face of the C ABI, used by the standalone encoder / decoder.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace dcvc {
namespace stream {

enum NalType : int { kSps = 0, kIntra = 1, kInter = 2 };

struct StreamError : std::runtime_error {
    int code;        // -1 range / space, -2 truncated, -3 malformed
    StreamError(int c, const char* what) : std::runtime_error(what), code(c) {}
};

// appends to `out`
void put_uint(std::vector<uint8_t>& out, uint32_t value);
void put_sps(std::vector<uint8_t>& out, int sps_id, int height, int width);
void put_ip(std::vector<uint8_t>& out, bool is_i, int sps_id, int qp, int ec_part, bool reset,
            const uint8_t* payload, size_t payload_bytes);

// sequential reader over caller memory
class Reader {
public:
    Reader(const uint8_t* p, size_t n) : m_p(p), m_n(n) {}
    bool at_end() const { return m_pos >= m_n; }
    size_t position() const { return m_pos; }
    void header(int& nal_type, int& sps_id);
    void sps_remaining(int& height, int& width);
    void ip_remaining(int& qp, int& ec_part, bool& reset, const uint8_t*& payload, size_t& payload_bytes);
    uint32_t get_uint();

private:
    const uint8_t* take(size_t k);
    const uint8_t* m_p;
    size_t m_n;
    size_t m_pos = 0;
};

// sequence parameter sets by picture size, ids 0..15 (SPSHelper, stream_helper.py:157-192)
class SpsTable {
public:
    struct Sps { int id, height, width; };
    // id of the set for this size; is_new = it has to be written to the stream first
    int id_for(int height, int width, bool& is_new);
    void add(int id, int height, int width);          // decoder side: replaces an existing id
    const Sps* find(int id) const;

private:
    std::vector<Sps> m_sets;
};

}  // namespace stream
}  // namespace dcvc
