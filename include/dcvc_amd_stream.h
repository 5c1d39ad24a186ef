/*
 * dcvc_amd_stream.h - C ABI of the bit-stream container of a coded sequence (libdcvc_amd.so).
 *
 * Replaces /root/reference/src/utils/stream_helper.py:37-154 (write_sps, write_ip, read_header,
 * read_sps_remaining, read_ip_remaining, write_uint_adaptive / read_uint_adaptive) for hosts that
 * do not run Python - the standalone encoder / decoder (dcvc_amd/bin/dcvc) is built on it. The
 * byte format is the reference's, byte for byte (tests/test_stream_native.py):
 *
 *   unit header   1 byte   nal_type << 4 | sps_id          nal_type: 0 SPS, 1 I picture, 2 P picture(s)
 *   SPS body      varuint height, varuint width
 *   I/P body      1 byte qp, 1 byte ec_parallel << 1 | reset_feature_memory, varuint length, payload
 *   varuint       < 2^7: 0vvvvvvv;  < 2^14: 10vvvvvv vvvvvvvv;  < 2^30: 11vvvvvv + 3 bytes  (big endian)
 *
 * Everything works on caller memory (no FILE*, no allocation): writers return the number of bytes
 * produced, readers the number consumed; a negative value is an error (dcvc_last_error()):
 * -1 destination too small / value out of range, -2 truncated input, -3 malformed input.
 */
#ifndef DCVC_AMD_STREAM_H
#define DCVC_AMD_STREAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* message of the last failing call on this thread (every dcvc_* entry point reports errors through it) */
#ifndef DCVC_LAST_ERROR_DECLARED
#define DCVC_LAST_ERROR_DECLARED
const char* dcvc_last_error(void);
#endif

#define DCVC_NAL_SPS 0
#define DCVC_NAL_I 1
#define DCVC_NAL_P 2

/* stream_helper.py:37-61 */
int dcvc_stream_write_uint(uint8_t* dst, size_t cap, uint32_t value);
int dcvc_stream_read_uint(const uint8_t* src, size_t n, uint32_t* value);

/* stream_helper.py:118-127 write_sps(f, sps) */
int dcvc_stream_write_sps(uint8_t* dst, size_t cap, int sps_id, int height, int width);
/* stream_helper.py:130-140 write_ip(f, is_i_frame, sps_id, qp, ec_part, reset_feature_memory, bit_stream);
 * dst == NULL returns the size the unit would take */
int64_t dcvc_stream_write_ip(uint8_t* dst, size_t cap, int is_i_frame, int sps_id, int qp, int ec_part,
                             int reset_feature_memory, const uint8_t* payload, size_t payload_bytes);

/* stream_helper.py:64-70 read_header(f) -> nal_type, sps_id */
int dcvc_stream_read_header(const uint8_t* src, size_t n, int* nal_type, int* sps_id);
/* stream_helper.py:73-79 read_sps_remaining(f, sps_id) -> height, width */
int dcvc_stream_read_sps_remaining(const uint8_t* src, size_t n, int* height, int* width);
/* stream_helper.py:143-154 read_ip_remaining(f) -> qp, ec_part, reset_feature_memory, bit_stream;
 * *payload points into src. Returns header + payload bytes consumed. */
int64_t dcvc_stream_read_ip_remaining(const uint8_t* src, size_t n, int* qp, int* ec_part,
                                      int* reset_feature_memory, const uint8_t** payload, size_t* payload_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DCVC_AMD_STREAM_H */
