/*
 * dcvc_amd_codec.h - C ABI of the DCVC-UF picture codecs (libdcvc_amd.so).
 *
 * Replaces the pybind11 module `inference_extensions_cuda`
 *   /root/reference/src/layers/extensions/inference/bind.cpp:11-39
 * (classes DMCIProxy / DMCHTSProxy / DMCHTLProxy / DMCLDProxy with set_param / compress /
 * decompress / add_ref_feature_from_frame). Tensors cross the boundary as plain pointers:
 * parameters in HOST memory, pictures as DEVICE pointers to fp16 NHWC ("channels_last") data on
 * the current HIP device; `stream` is a hipStream_t. Returns 0 (or the documented value) on
 * success, a negative value on error (message in dcvc_last_error()).
 */
#ifndef DCVC_AMD_CODEC_H
#define DCVC_AMD_CODEC_H

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

/* Version of the arithmetic policy this build computes with (contraction order and rounding points, the WSiLU
 * evaluation, the symbol kernels' fp16 steps). Streams decode bit-exactly only between builds of the SAME version: the
 * container (the reference's own, stream_helper.py:130-154) has no field for it, so callers that keep streams around record
 * it beside them - the native tool writes it into its JSON logs. Round 2: 2, rounds 3 and 4: 3. No reference counterpart (the
 * reference makes no cross-build guarantee). */
int dcvc_arith_policy_version(void);

typedef struct dcvc_dmci dcvc_dmci;

/* tensor element types of dcvc_*_set_param */
#define DCVC_F16 0
#define DCVC_F32 1
#define DCVC_I32 2

/* bind.cpp:12-16  DMCIProxy() */
dcvc_dmci* dcvc_dmci_create(void);
void dcvc_dmci_destroy(dcvc_dmci* c);

/* DMCIProxy.set_param(state_dict, skip_thres), dmci_proxy.cpp:604-652.
 * n tensors: names[i], data[i] (host memory), dtypes[i] (DCVC_*), ndims[i], and their dims
 * concatenated in `dims`. The state_dict is the module's state_dict() plus the four int32 CDF
 * tensors gaussian_encoder.{quantized_cdf,cdf_length}, bit_estimator_z.{quantized_cdf,cdf_length}
 * (common_model.py:64-70). Everything is copied. */
int dcvc_dmci_set_param(dcvc_dmci* c, int n, const char* const* names, const void* const* data,
                        const int* dtypes, const int* ndims, const int64_t* dims, float skip_thres);

/* DMCIProxy.compress(x, qp, padding_b, padding_r) -> (bit_stream, x_hat, ec_parallel),
 * dmci_proxy.cpp:296-421. x: device fp16 [height][width][3] in [-0.5, 0.5], unpadded;
 * x_hat: device fp16 [ceil16(height)][ceil16(width)][3], written on `stream` (the call returns
 * when the bit stream is ready; reconstruction kernels may still be in flight, as in the
 * reference). padding_b / padding_r must equal the padding to a multiple of 16.
 * Returns ec_parallel (1..8); the bytes are fetched with dcvc_dmci_get_stream. */
int dcvc_dmci_compress(dcvc_dmci* c, const void* x, int height, int width, int qp, int padding_b,
                       int padding_r, void* x_hat, void* stream);
/* size query (dst == NULL) or copy of at most cap bytes; returns the stream size */
int64_t dcvc_dmci_get_stream(dcvc_dmci* c, uint8_t* dst, size_t cap);

/* DMCIProxy.decompress(bit_stream, qp, height, width, ec_parallel) -> x_hat,
 * dmci_proxy.cpp:423-602. */
int dcvc_dmci_decompress(dcvc_dmci* c, const uint8_t* bit_stream, size_t nbytes, int qp, int height,
                         int width, int ec_parallel, void* x_hat, void* stream);

/* Not part of the reference surface: 1 = replay the stages as hipGraphs (default), 0 = eager. */
int dcvc_dmci_set_use_graphs(dcvc_dmci* c, int on);
/* Test hook: copy an internal tensor of the last call ("y", "y_hat", "z_i8", "params",
 * "unshuffled", "features", "totals", "symbols") to host memory; returns its size in bytes. */
int64_t dcvc_dmci_debug_read(dcvc_dmci* c, const char* name, void* dst, size_t cap, void* stream);

/* ------------------------------------------------------------------ DMCLDProxy (low-delay inter)
 * bind.cpp:31-38 / dmc_ld_proxy.h. One picture per call; the temporal state (feature memory,
 * last decoded feature, context, temporal prior) stays on the device inside the object. */
typedef struct dcvc_dmcld dcvc_dmcld;

dcvc_dmcld* dcvc_dmcld_create(void);
void dcvc_dmcld_destroy(dcvc_dmcld* c);

/* DMCLDProxy.set_param(state_dict, skip_thres), dmc_ld_proxy.cpp:595-639; arguments as for
 * dcvc_dmci_set_param. Clears the temporal state. */
int dcvc_dmcld_set_param(dcvc_dmcld* c, int n, const char* const* names, const void* const* data,
                         const int* dtypes, const int* ndims, const int64_t* dims, float skip_thres);

/* DMCLDProxy.add_ref_feature_from_frame(frame, apply_feature_adaptor), dmc_ld_proxy.cpp:407-418.
 * frame: device fp16 [height][width][3], the intra codec's reconstruction. apply_adaptor != 0 is
 * the encoder's call (memory, context and temporal prior are derived immediately), 0 the
 * decoder's (they are derived by the next decompress). */
int dcvc_dmcld_add_ref_feature_from_frame(dcvc_dmcld* c, const void* frame, int height, int width,
                                          int apply_adaptor, void* stream);

/* DMCLDProxy.compress(x, qp, reset_feature_memory, padding_b, padding_r) -> (bit_stream,
 * ec_parallel), dmc_ld_proxy.cpp:420-473. Returns ec_parallel; bytes via dcvc_dmcld_get_stream. */
int dcvc_dmcld_compress(dcvc_dmcld* c, const void* x, int height, int width, int qp,
                        int reset_feature_memory, int padding_b, int padding_r, void* stream);
int64_t dcvc_dmcld_get_stream(dcvc_dmcld* c, uint8_t* dst, size_t cap);

/* DMCLDProxy.decompress(bit_stream, qp, height, width, ec_parallel, reset_feature_memory) ->
 * x_hat, dmc_ld_proxy.cpp:475-593. x_hat: device fp16 [ceil16(height)][ceil16(width)][3]. */
int dcvc_dmcld_decompress(dcvc_dmcld* c, const uint8_t* bit_stream, size_t nbytes, int qp, int height,
                          int width, int ec_parallel, int reset_feature_memory, void* x_hat,
                          void* stream);

/* Not part of the reference surface - hand-off of a GOP to another GPU (north_star: temporal
 * context exchanged point-to-point over xGMI): the temporal state (reference feature, memory,
 * last decoded feature, context, temporal prior, validity flags) as ONE flat device buffer that
 * torch.distributed.send / recv (RCCL) can move. export with dst == NULL returns the size in bytes;
 * import needs a codec with the same parameters and picture size. */
int64_t dcvc_dmcld_export_state(dcvc_dmcld* c, void* dst, size_t cap, void* stream);
int dcvc_dmcld_import_state(dcvc_dmcld* c, const void* src, size_t bytes, int height, int width, void* stream);

int dcvc_dmcld_set_use_graphs(dcvc_dmcld* c, int on);
/* Test hook ("y", "y_hat", "common", "means1", "z_i8", "memory", "feature_p", "ctx", "temporal",
 * "feature_i", "symbols", "totals"); dense copy, returns the size in bytes. */
int64_t dcvc_dmcld_debug_read(dcvc_dmcld* c, const char* name, void* dst, size_t cap, void* stream);

/* ------------------------------------------------------------------ DMCHTSProxy / DMCHTLProxy
 * bind.cpp:17-30 / dmc_hts_proxy.h, dmc_htl_proxy.h: the hierarchical inter codecs, 8 pictures
 * ("chunk") per call. is_hts != 0 creates the HT-S codec, 0 the HT-L codec; set_param rejects a
 * state_dict of the other structure. */
typedef struct dcvc_dmcht dcvc_dmcht;

dcvc_dmcht* dcvc_dmcht_create(int is_hts);
void dcvc_dmcht_destroy(dcvc_dmcht* c);
int dcvc_dmcht_set_param(dcvc_dmcht* c, int n, const char* const* names, const void* const* data,
                         const int* dtypes, const int* ndims, const int64_t* dims, float skip_thres);
/* dmc_hts_proxy.cpp:492-502 */
int dcvc_dmcht_add_ref_feature_from_frame(dcvc_dmcht* c, const void* frame, int height, int width,
                                          int apply_adaptor, void* stream);
/* DMCHT*Proxy.compress(x, qp, reset_feature_memory, padding_b, padding_r), dmc_hts_proxy.cpp:504-585.
 * x: device fp16 [height][width][24] = the channels_last memory of the reference's [1, 24, H, W]
 * input (8 pictures x 3 planes). Returns ec_parallel. */
int dcvc_dmcht_compress(dcvc_dmcht* c, const void* x, int height, int width, int qp,
                        int reset_feature_memory, int padding_b, int padding_r, void* stream);
int64_t dcvc_dmcht_get_stream(dcvc_dmcht* c, uint8_t* dst, size_t cap);
/* DMCHT*Proxy.decompress(...) -> list of 8 x_hat, dmc_hts_proxy.cpp:587-710. x_hat: device fp16
 * [8][ceil16(height)][ceil16(width)][3], the 8 reconstructions back to back. */
int dcvc_dmcht_decompress(dcvc_dmcht* c, const uint8_t* bit_stream, size_t nbytes, int qp, int height,
                          int width, int ec_parallel, int reset_feature_memory, void* x_hat,
                          void* stream);
/* GOP hand-off between GPUs, as dcvc_dmcld_export_state / _import_state */
int64_t dcvc_dmcht_export_state(dcvc_dmcht* c, void* dst, size_t cap, void* stream);
int dcvc_dmcht_import_state(dcvc_dmcht* c, const void* src, size_t bytes, int height, int width, void* stream);
/* Not part of the reference surface - reconstruction-head fan-out over several GPUs (SURVEY 8e iii;
 * video_model_ht.py:252-275: the 8 picture heads depend only on feature_p). The GPU that holds the
 * stream decodes with dcvc_dmcht_set_recon_mask(own pictures), exports feature_p (dense device
 * buffer [P8][512] fp16; dst == NULL returns the size), the others import it and run their heads:
 * x_hat + i * picture is written for every picture i of `mask` (bit i). Bit 7 (the last picture)
 * must stay with the GPU that keeps the temporal state: its head output is the reset feature. */
int dcvc_dmcht_set_recon_mask(dcvc_dmcht* c, unsigned mask);
int64_t dcvc_dmcht_export_feature(dcvc_dmcht* c, void* dst, size_t cap, void* stream);
int dcvc_dmcht_import_feature(dcvc_dmcht* c, const void* src, size_t bytes, int height, int width, void* stream);
int dcvc_dmcht_run_recon_heads(dcvc_dmcht* c, unsigned mask, void* x_hat, void* stream);
int dcvc_dmcht_set_use_graphs(dcvc_dmcht* c, int on);
/* Test hook ("y", "y_hat", "common", "z_i8", "memory", "feature_p", "ctx", "feature_i", "symbols",
 * "totals"). */
int64_t dcvc_dmcht_debug_read(dcvc_dmcht* c, const char* name, void* dst, size_t cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCVC_AMD_CODEC_H */
