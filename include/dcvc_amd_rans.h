/*
 * dcvc_amd_rans.h - C ABI of the host rANS entropy coder (libdcvc_amd.so).
 *
 * Replaces the pybind11 module `MLCodec_extensions_cpp`
 *   /root/reference/src/cpp/py_rans/bind.cpp:14-40
 * (classes RansEncoder / RansDecoder and pmf_to_quantized_cdf). One entry point per bound
 * method; arrays are plain pointers + counts. All functions return 0 on success and a negative
 * value on error (dcvc_last_error() then holds the message), unless stated otherwise.
 */
#ifndef DCVC_AMD_RANS_H
#define DCVC_AMD_RANS_H

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

typedef struct dcvc_rans_encoder dcvc_rans_encoder;
typedef struct dcvc_rans_decoder dcvc_rans_decoder;

/* Message of the last failed call on this thread ("" if none). */
const char* dcvc_last_error(void);

/* bind.cpp:40  pmf_to_quantized_cdf(list[float]) -> list[uint32]; cdf_out has n + 1 entries. */
int dcvc_pmf_to_quantized_cdf(const float* pmf, int n, uint32_t* cdf_out);

/* bind.cpp:16-27  RansEncoder */
dcvc_rans_encoder* dcvc_rans_encoder_create(void);
void dcvc_rans_encoder_destroy(dcvc_rans_encoder* e);
/* set_cdf(cdfs int32[num_cdf, stride], cdf_sizes int32[num_cdf], index in {0: z, 1: y}) */
int dcvc_rans_encoder_set_cdf(dcvc_rans_encoder* e, const int32_t* cdfs, int num_cdf, int stride,
                              const int32_t* cdf_sizes, int index);
int dcvc_rans_encoder_set_entropy_coder_parallel(dcvc_rans_encoder* e, int n);
int dcvc_rans_encoder_reset(dcvc_rans_encoder* e);
/* encode_y(int16[count]): each entry is (symbol << 8) + cdf_index. The array is copied. */
int dcvc_rans_encoder_encode_y(dcvc_rans_encoder* e, const int16_t* symbols, int count);
/* encode_z(int8[count], cdf_offset, ch): cdf index of entry i is (i % ch) + cdf_offset. Copied. */
int dcvc_rans_encoder_encode_z(dcvc_rans_encoder* e, const int8_t* symbols, int count,
                               int cdf_offset, int ch);
int dcvc_rans_encoder_flush(dcvc_rans_encoder* e);
/* get_encoded_stream(): size query (dst == NULL) or copy of at most cap bytes; returns size. */
int64_t dcvc_rans_encoder_get_encoded_stream(dcvc_rans_encoder* e, uint8_t* dst, size_t cap);

/* bind.cpp:29-38  RansDecoder */
dcvc_rans_decoder* dcvc_rans_decoder_create(void);
void dcvc_rans_decoder_destroy(dcvc_rans_decoder* d);
int dcvc_rans_decoder_set_cdf(dcvc_rans_decoder* d, const int32_t* cdfs, int num_cdf, int stride,
                              const int32_t* cdf_sizes, int index);
int dcvc_rans_decoder_set_entropy_coder_parallel(dcvc_rans_decoder* d, int n);
int dcvc_rans_decoder_set_stream(dcvc_rans_decoder* d, const uint8_t* data, size_t size);
/* decode_y(uint8 indexes[count]) -> int8 out[count] (the reference keeps the result inside the
 * object, py_rans.h:60; here the caller provides the output array). */
int dcvc_rans_decoder_decode_y(dcvc_rans_decoder* d, const uint8_t* indexes, int count,
                               int8_t* out);
int dcvc_rans_decoder_decode_z(dcvc_rans_decoder* d, int count, int cdf_offset, int ch,
                               int8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* DCVC_AMD_RANS_H */
