/*
 * dcvc_amd_ops.h - C ABI of the individual HIP kernels (libdcvc_amd.so).
 *
 * One entry point per device function the reference's proxies call
 *   /root/reference/src/layers/extensions/inference/def_cutlass.h:10-40     (fused conv ops)
 *   /root/reference/src/layers/extensions/inference/def_elementwise.h:10-78 (elementwise ops)
 * with at::Tensor replaced by (device pointer, leading dimension, sizes). All tensors are fp16
 * NHWC ("channels_last") on the current HIP device; `stream` is a hipStream_t (0 = default).
 * Nothing allocates or synchronises. Returns 0, or -1 with dcvc_last_error() set.
 */
#ifndef DCVC_AMD_OPS_H
#define DCVC_AMD_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* message of the last failing call on this thread (every dcvc_* entry point reports errors through it) */
#ifndef DCVC_LAST_ERROR_DECLARED
#define DCVC_LAST_ERROR_DECLARED
const char* dcvc_last_error(void);
#endif

/* flags for dcvc_conv1x1 */
#define DCVC_CONV_WSILU      1   /* y = wsilu(acc + bias)                     (conv1x1_bias_wsilu)            */
#define DCVC_CONV_CHUNK_ADD  2   /* sum of 4 adjacent output channels         (conv1x1_bias_wsilu_chunk_add)  */

/* def_cutlass.h:10-33: conv1x1_bias / _wsilu / _shortcut / _shortcut2 / _shortcut_with_quant /
 * _with_quant / _wsilu_chunk_add, selected by which optional pointers are non-NULL and `flags`.
 *   y[p][n] = fp16( ((acc + bias[n]) (wsilu) + r1[p][n] + r2[p][n]) * q[n] ), then * q2[n] in fp16.
 * x: [pixels][ldx] (first cin channels), w: [cout][cin], y: [pixels][ldy]. */
int dcvc_conv1x1(const void* x, int ldx, const void* w, const void* bias,
                 const void* r1, int ldr1, const void* r2, int ldr2,
                 const void* q, const void* q2, void* y, int ldy,
                 int pixels, int cin, int cout, int flags, void* stream);

/* def_cutlass.h:35-37 conv_bias: dense k x k conv (k in {2,3}), stride in {1,2}, zero padding.
 * w: [cout][k][k][cin] (tap-major re-layout of the PyTorch [cout][cin][k][k] weight). */
int dcvc_conv_kxk(const void* x, int ldx, const void* w, const void* bias, void* y, int ldy,
                  int in_h, int in_w, int cin, int cout, int ksize, int stride, int pad,
                  void* stream);

/* def_cutlass.h:39-40 transposed_conv: 2x2, stride 2, no bias. w: [4 = dy*2+dx][cout][cin]. */
int dcvc_tconv2x2(const void* x, int ldx, const void* w, void* y, int ldy,
                  int in_h, int in_w, int cin, int cout, void* stream);

/* def_cutlass.h:34 d3x3: depthwise 3x3, pad 1, no bias. w: [9][C]. */
int dcvc_dwconv3x3(const void* x, int ldx, const void* w, void* y, int ldy, int H, int W, int C,
                   void* stream);

/* def_elementwise.h: pad_and_unshuffle_8_cuda / pixel_shuffle_8_cuda / pixel_shuffle_2_cuda /
 * replicate_pad_cuda / slice_cuda / multiply_with_broadcast_cuda */
int dcvc_pad_unshuffle8(const void* x, int H, int W, int C3, void* out, int H8, int W8, void* stream);
int dcvc_shuffle8(const void* in, int ldin, int H8, int W8, int C3, int clamp, void* out, void* stream);
int dcvc_shuffle2(const void* in, int ldin, int H, int W, int C, void* out, int ldout, void* stream);
int dcvc_replicate_pad(const void* in, int ldin, int H, int W, int C, int pad_b, int pad_r,
                       void* out, int ldout, void* stream);
int dcvc_crop(const void* in, int ldin, int Win, void* out, int ldout, int H, int W, int C, void* stream);
int dcvc_mul_channel(const void* x, int ldx, const void* q, void* y, int ldy, int pixels, int C,
                     void* stream);

/* ffn.0 + ffn.2 of a DepthConvBlock in one launch (layers_proxy.cpp:84-98: conv1x1_bias_wsilu_chunk_add
 * followed by conv1x1_bias_shortcut[2][_with_quant] with the block-internal tensor as first residual):
 *   out = W2 * chunk_add(WSiLU(W0 * x + b0)) + b2 + x [+ r2] [* q], rounded to fp16, [* q2].
 * c in {128, 256, 384}, cffn a multiple of 64; bit-identical to the two-launch sequence; y may alias x. */
int dcvc_ffn_fused(const void* x, int ldx, const void* w0, const void* b0, const void* w2, const void* b2,
                   const void* r2, int ldr2, const void* q, const void* q2, void* y, int ldy,
                   int pixels, int c, int cffn, void* stream);

/* A DepthConvBlock behind its depthwise conv in ONE launch (layers_proxy.cpp:81-98: conv1x1_bias_shortcut,
 * conv1x1_bias_wsilu_chunk_add, conv1x1_bias_shortcut[2][_with_quant]), optionally followed by dc.0 of the NEXT block of a
 * chain (conv1x1_bias_wsilu, layers_proxy.cpp:79):
 *   y1 = W3 * t2 + b3 + x;  y = (W2 * chunk_add(WSiLU(W0 * y1 + b0)) + b2 + y1 [+ x]) [* q] -> fp16 [* q2];
 *   t1n = WSiLU(W1n * y + b1n)   (when w1n != NULL).
 * t2 = depthwise output [pixels][ldt], x = block input [pixels][ldx], b3 = dc.3 bias with the depthwise bias folded in,
 * ci = inner width (cdc = cffn). Bit-identical to the separate launches; y may alias x when !shortcut.
 * The N-split kernel (round 3: activations in LDS, every wave owns a share
 * of the output channels and streams its weight fragments from a packed copy of w3 | w0 | w2 [| w1n] that this entry point
 * builds on EVERY call in stream-ordered temporaries - the codecs pack once at set_param time; callers that launch many
 * times use the handle form below).
 * (c, ci) in {(256, 256), (384, 384), (512, 512), (768, 768)} - full-width blocks - and {(512, 256), (256, 128)}: the
 * half-width `dcb2` blocks of the inter models (layers.py:128-159; w3 [c][ci], w0 [4 ci][c], w2 [c][ci], w1n [ci][c]).
 * Bit-identical to dcvc_dcb_tail and to the separate launches. */
int dcvc_dcb_nsplit(const void* t2, int ldt, const void* x, int ldx, const void* w3, const void* b3,
                    const void* w0, const void* b0, const void* w2, const void* b2, const void* q, const void* q2,
                    const void* w1n, const void* b1n, void* t1n, int ldt1, void* y, int ldy,
                    int pixels, int c, int ci, int shortcut, void* stream);
/* The same block as the LAST one of a chain, with the 1x1 conv that closes the chain inside the launch (round 6; the reference
 * launches conv1x1_bias / conv1x1_bias_with_quant behind the block: y_prior_fusion.conv.3 dmci_proxy.cpp:172-176,
 * y_spatial_prior.conv.3 dmci_proxy.cpp:196-199, decoder.conv2 dmc_ld_proxy.cpp:484-487, recon_head.head :499-503):
 *   yfin = (Wfin * y + bfin) [* qfin] -> fp16,  Wfin [nfin][c].
 * Returns an error for an (c, ci, nfin) the kernel has no variant of (dcvc_dcb_nsplit_fin_supported). Bit-identical to
 * dcvc_dcb_nsplit followed by dcvc_conv1x1(bias [, q]). */
int dcvc_dcb_nsplit_fin(const void* t2, int ldt, const void* x, int ldx, const void* w3, const void* b3,
                        const void* w0, const void* b0, const void* w2, const void* b2, const void* q, const void* q2,
                        const void* wfin, const void* bfin, const void* qfin, void* yfin, int ldyfin, int nfin,
                        void* y, int ldy, int pixels, int c, int ci, int shortcut, void* stream);
int dcvc_dcb_nsplit_fin_supported(int c, int ci, int nfin);
/* The same block WITH its depthwise 3x3 conv inside the launch (round 6; the reference launches d3x3 between dc.0 and dc.3,
 * layers_proxy.cpp:80-84, cutlass/d3x3.cu:443-446): t1 [pixels][ldt] = dc.0's output (what dcvc_dwconv3x3 would read), wdw = the
 * taps [9][ci] (tap major, as dcvc_dwconv3x3 takes them), width = the picture's width (pixels = rows x width, row-major). Behind
 * ffn.2 either dc.0 of the next block (w1n / b1n / t1n; t1n must not be t1) or a chain's closing conv (wfin ... nfin) or neither.
 * Returns an error for a (c, ci, pixels) the kernel has no such variant of (dcvc_dcb_nsplit_dw_supported: (256, 128); (384, 192) below
 * 12 800 pixels, where the workgroups take 32 pixels and LDS has room for the rows around them). Bit-identical to
 * dcvc_dwconv3x3 followed by dcvc_dcb_nsplit / dcvc_dcb_nsplit_fin. */
int dcvc_dcb_nsplit_dw(const void* t1, int ldt, const void* wdw, int width, const void* x, int ldx, const void* w3, const void* b3,
                       const void* w0, const void* b0, const void* w2, const void* b2, const void* q, const void* q2,
                       const void* w1n, const void* b1n, void* t1n, int ldt1,
                       const void* wfin, const void* bfin, const void* qfin, void* yfin, int ldyfin, int nfin,
                       void* y, int ldy, int pixels, int c, int ci, int shortcut, void* stream);
int dcvc_dcb_nsplit_dw_supported(int c, int ci, int pixels);

/* The two 1x1 convs in FRONT of a block's depthwise conv in one launch (round 6; the reference launches conv1x1_bias for the
 * adaptor, layers_proxy.cpp:73-77, then conv1x1_bias_wsilu for dc.0, :79):
 *   y = Wa * x + ba  (Wa [c][cin]);   t1 = WSiLU(W1 * y + b1)  (W1 [ci][c]).
 * Returns an error for a (cin, c, ci) the kernel has no variant of (dcvc_dcb_pair_supported). y must not alias x.
 * Bit-identical to dcvc_conv1x1(bias) followed by dcvc_conv1x1(bias, wsilu). */
int dcvc_dcb_pair(const void* x, int ldx, const void* wa, const void* ba, const void* w1, const void* b1,
                  void* y, int ldy, void* t1, int ldt1, int pixels, int cin, int c, int ci, void* stream);
int dcvc_dcb_pair_supported(int cin, int c, int ci);

/* Handle form: pack w3 | w0 | w2 (and w1n, or NULL) once - the packed copies are a snapshot of the weights at pack time -,
 * launch any number of times, free (synchronises the device the handle was packed on, whichever is current). `stream` of
 * _pack and of _packed may differ: _packed orders its stream behind the pack launches (an event recorded by _pack; only until
 * the event has been seen complete, and never on the pack stream itself). A stream that is being CAPTURED into a hipGraph must not
 * be the first to launch with a handle packed on another stream: pack and first launch belong in front of the capture.
 * with_next != 0 runs the next block's dc.0 inside the launch (the handle must have been packed with w1n). No reference counterpart: the reference's CUTLASS kernels read the
 * row-major matrices directly. */
int dcvc_dcb_nsplit_pack(const void* w3, const void* w0, const void* w2, const void* w1n, int c, int ci, void* stream,
                         void** handle);
int dcvc_dcb_nsplit_packed(const void* handle, const void* t2, int ldt, const void* x, int ldx, const void* b3,
                           const void* b0, const void* b2, const void* q, const void* q2, const void* b1n,
                           void* t1n, int ldt1, void* y, int ldy, int pixels, int shortcut, int with_next, void* stream);
int dcvc_dcb_nsplit_free(void* handle);

/* DepthConvBlockProxy::forward behind dc.0 (layers_proxy.cpp:79-98: d3x3, conv1x1_bias_shortcut,
 * conv1x1_bias_wsilu_chunk_add, conv1x1_bias_shortcut[2][_with_quant]) in one launch for the
 * half-width blocks of the inter models: t = dc.0 output [H*W][ldt]; dw = [9][cdc] tap-major depthwise
 * weights (NULL: t already is the depthwise output); x = block-internal input (residual of dc.3, and
 * of ffn.2 when shortcut != 0); c in {128, 256}, cdc <= 128, cffn: multiples of 64. Bit-identical to
 * the four-launch sequence; y may alias x. With w1 / b1 (dc.0 weights [cdc][c] and bias) non-NULL, dc.0
 * (conv1x1_bias_wsilu on x) runs inside the launch too and t is not read - the whole block behind an
 * optional adaptor in one launch; y must then NOT alias x (patches read their neighbours' input). */
int dcvc_dcb_tail(const void* w1, const void* b1, const void* t, int ldt, const void* dw, const void* x, int ldx,
                  const void* w3, const void* b3,
                  const void* w0, const void* b0, const void* w2, const void* b2, const void* q, const void* q2,
                  void* y, int ldy, int H, int W, int c, int cdc, int cffn, int shortcut, void* stream);

/* Debugging aid (no reference counterpart): device buffer [H*W][cdc] that receives dc.0's output from
 * the following dcvc_dcb_tail launches with dc.0 inside; NULL = off. */
int dcvc_dcb_tail_debug_buffer(void* device_buffer);

/* stream.cu:40-76 / 422-443: y = x * max(q, 0.5) or, with reciprocal != 0, y = x * fp16(1 / max(q, 0.5));
 * q is a tensor of the same shape (the inter models' per-element quantisation step). */
int dcvc_scale_clamped(const void* x, int ldx, const void* q, int ldq, void* y, int ldy, int pixels,
                       int C, int reciprocal, void* stream);

/* Picture I/O on the device, replacing the host-side numpy/scipy/torch chains of the harness.
 * test_video.py:69-123 get_src_frame (+ transforms.py:69-80 ycbcr420_to_444_np, order 0):
 *   y: u8 [H][W], uv: u8 [2][H/2][W/2] (device) -> x fp16 at pixel stride ldx (3 channels written):
 *   nearest chroma, x = fp16(fp16(v / 255) - 0.5). */
int dcvc_yuv420_to_x(const void* y, const void* uv, int H, int W, void* x, int ldx, void* stream);
/* test_video.py:32-45 get_distortion and :356-363 (decoded-picture writer):
 *   x_hat fp16 [rows][row_pixels][3] -> top-left H x W picture; y16/uv16: fp16 planes in 0..255,
 *   y8/uv8: u8 planes (Y rounded half-to-even, U/V truncated as the reference does). Null = skip. */
int dcvc_x_to_yuv420(const void* x_hat, int row_pixels, int H, int W, void* y16, void* uv16, void* y8,
                     void* uv8, void* stream);

/* Tuning aid (no reference counterpart): device buffer of [blocks][16] int64 shader-clock stamps
 * written by wave 0 of every workgroup of the following contraction launches; NULL = off. */
int dcvc_gemm_timeline_buffer(void* device_buffer);
int dcvc_dcb_nsplit_timeline_buffer(void* device_buffer);   /* [workgroups][32] stamps of the N-split block kernel */
/* tuning aid (tools/probes/core_bench.hip -w): while set, dcvc_dcb_nsplit* launches of a shape that has the variant run with their
 * depthwise conv inside on these operands (t1 [pixels][ci] instead of the call's t2, taps [9][ci], picture width); t1 = NULL: off */
int dcvc_dcb_nsplit_dw_hook(const void* t1, const void* wdw, int width);

/* def_elementwise.h: round_z_cuda / int8_to_dtype_cuda */
int dcvc_round_z(const void* z, void* z_hat, void* z_i8, int count, void* stream);
int dcvc_int8_to_half(const void* in, void* out, int count, void* stream);

/* One autoregressive step of the 4x masked y coding, encoder side. Fuses
 * process_with_mask_cuda + single_part_for_writing_4x_cuda (x2) + build_index_enc_cuda +
 * conditional_index_part1_cuda (def_elementwise.h). Outputs: y_hat_acc (active group written;
 * step 0 also zeroes the other groups), sym[P*C/4] int16, cond bits, per-block counts, then the
 * compacted symbols out[...] and totals[step]. n_blocks = dcvc_symbol_blocks(P*C/4). */
int dcvc_symbol_blocks(int count);
int dcvc_y_step_enc(const void* y, int ldy, const void* scales, int lds, const void* means, int ldm,
                    void* y_hat_acc, int ldacc, void* sym, void* cond, void* block_count,
                    void* compact_out, void* totals,
                    int H, int W, int C, int step, float skip_thres, void* stream);
/* decoder side, part 1: single_part_for_reading_4x_cuda + build_index_dec_cuda + compaction */
int dcvc_y_step_dec_index(const void* scales, int lds, void* index, void* cond, void* block_count,
                          void* compact_out, void* totals,
                          int H, int W, int C, int step, float skip_thres, void* stream);
/* decoder side, part 2: conditional_recover_with_type_conversion_cuda + restore_y_4x*_cuda.
 * decoded: int8 symbols of ALL steps so far, this step's start at sum(totals[0..step)). */
int dcvc_y_step_dec_restore(const void* decoded, const void* cond, const void* block_count,
                            const void* totals, const void* means, int ldm,
                            void* y_hat_acc, int ldacc, int H, int W, int C, int step, void* stream);

/* The inter models' full-tensor masked steps (nsteps = 2: LD checkerboard x channel halves,
 * dmc_ld_proxy.cpp:672-683; nsteps = 4: HT-S channel-group x 2x2-position masks,
 * dmc_hts_proxy.cpp:869-890). One call = one step of
 *   divide_with_clamp_min_inplace_cuda (step 0) + process_with_mask_no_scale[_add[_and_multiply]]_inplace_cuda
 *   (+ build_index_enc_cuda + conditional_index_part1_cuda on the last step, compacted symbols -> compact_out,
 *   count -> totals[0]).
 * y is rescaled in place at step 0; y_hat accumulates over the steps and is final after the last. */
int dcvc_mask_step_enc(void* y, int ldy, const void* q_dec, int ldq, const void* scales, int lds,
                       const void* means, int ldm, void* y_hat, int ldh, void* sym, void* cond,
                       void* block_count, void* compact_out, void* totals, int H, int W, int C,
                       int nsteps, int step, float skip_thres, void* stream);
/* build_index_dec_cuda + conditional_index_part1_cuda over all channels */
int dcvc_mask_dec_index(const void* scales, int lds, void* index, void* cond, void* block_count,
                        void* compact_out, void* totals, int H, int W, int C, float skip_thres, void* stream);
/* conditional_recover_with_type_conversion_cuda (step 0) + restore_y[_and_add[_multiply]]_inplace_cuda;
 * yq: int8 scratch [H*W*C] carried from step 0 to the later steps */
int dcvc_mask_step_dec(const void* decoded, const void* cond, const void* block_count, const void* totals,
                       void* yq, const void* means, int ldm, const void* q_dec, int ldq, void* y_hat, int ldh,
                       int H, int W, int C, int nsteps, int step, void* stream);

/* Measurement hook (bench.py roofline leg, not a reference entry point): bracket every
 * contraction launch with HIP events on its stream; collect = summed kernel milliseconds,
 * algorithmic FLOPs (2*M*N*K) and launch count since the last reset. Graph replay must be off. */
int dcvc_gemm_profile_enable(int on);
int dcvc_gemm_profile_reset(void);
int dcvc_gemm_profile_collect(double* ms, double* flops, long long* launches);
/* per-launch records {int M, N, K, variant; float ms}; returns the number of launches recorded */
long long dcvc_gemm_profile_launches(void* records, long long cap);

#ifdef __cplusplus
}
#endif
#endif /* DCVC_AMD_OPS_H */
