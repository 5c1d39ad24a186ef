// Host-side launch interface of the HIP kernels (raw device pointers + explicit leading
// dimensions, everything on the caller's stream; nothing here allocates or synchronises, so all
// of it can be captured into a hipGraph).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>

namespace dcvc {

typedef _Float16 half_t;

inline void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) {
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }
}

// One-time device-side tables (WSiLU coefficients, scale->index LUT). Call once per process
// before the first launch and outside any graph capture.
void kernels_init();

// ---------------------------------------------------------------- per-kernel timing (conv_gemm.hip)
// When enabled, every contraction launch is bracketed by a pair of HIP events on its own stream
// (launches must then be eager, not captured). collect() synchronises the events and returns
// the summed kernel time, the algorithmic FLOPs (2*M*N*K) and the launch count since reset().
void gemm_profile_enable(bool on);
void gemm_profile_reset();
void gemm_profile_collect(double* ms, double* flops, long long* launches);
struct GemmLaunchInfo {
    int M, N, K, variant;   // variant bits: 1 spatial, 2 wsilu, 4 chunk-add, 8/16 residuals, 32 quant, 64 upsample
    float ms;
};
size_t gemm_profile_launches(GemmLaunchInfo* out, size_t cap);
// Other contraction kernels (dcb_nsplit.hip, dcb_tail.hip, ffn_fused.hip) register their launches in the same list: when profiling
// is on, reserves a record and hands out the two events hipExtLaunchKernelGGL stamps; false = off.
// info.K is chosen such that 2 * M * N * K is the launch's FLOP count; variant bits 28..31 name the kernel
// family: 0 conv_gemm, 2 dcb_tail, 3 ffn_fused, 4 dcb_nsplit, 5 dcb_nsplit8 (its variant also carries the inner width in bits
// 0..11 and the NEXT slot - 0, 1 = next dc.0, NN = closing conv - in bits 12..23), 6 dcb_pair8 (8 was round 2's dcb_core).
bool gemm_profile_slot(const GemmLaunchInfo& info, hipEvent_t* start, hipEvent_t* stop);
// Tuning aid: when non-null, wave 0 of every workgroup of the following contraction launches
// writes up to 16 shader-clock stamps (kernel entry, prologue issued, start of k-steps 0..7, main
// loop done, epilogue math done, stores issued) to buffer[block * 16 + i]. Null switches it off.
void gemm_timeline_buffer(long long* device_buffer);

// ---------------------------------------------------------------- dense convolutions (conv_gemm.hip)
struct Conv1x1Desc {
    const half_t* x = nullptr; int ldx = 0;     // [pixels][ldx], first `cin` channels of each pixel
    const half_t* w = nullptr;                  // [cout][cin]
    const half_t* bias = nullptr;               // [cout] or null
    const half_t* r1 = nullptr; int ldr1 = 0;   // residuals (added in fp32 before the rounding)
    const half_t* r2 = nullptr; int ldr2 = 0;
    const half_t* q = nullptr;                  // fused per-channel scale (…_with_quant)
    const half_t* q2 = nullptr;                 // per-channel scale applied to the rounded fp16 output
    half_t* y = nullptr; int ldy = 0;           // [pixels][ldy]; cout (or cout/4 with chunk_add) channels
    int pixels = 0, cin = 0, cout = 0;
    bool wsilu = false, chunk_add = false;
};
void conv1x1(const Conv1x1Desc& d, hipStream_t stream);

// DepthConvBlock behind its depthwise conv in one launch, "N-split" form (dcb_nsplit.hip, round 3; round 2's register-resident
// dcb_core kernel is in the history only):
//   y1 = W3 t2 + b3 + x ; t = chunk_add(WSiLU(W0 y1 + b0)) ; y = (W2 t + b2 + y1 [+ x]) [* q] -> fp16 [* q2]
//   and optionally the next block's dc.0: t1n = WSiLU(W1n y + b1n).
// Bit-identical to conv1x1(dc.3) + conv1x1(ffn.0, wsilu, chunk_add) + conv1x1(ffn.2) [+ conv1x1(dc.0, wsilu)].
// Activations in LDS, every wave owns a quarter of
// the output channels and streams ITS weight fragments straight from L2 out of a pre-packed per-wave stream
// (dcb_nsplit_pack_main / _dc0, packed once at set_param time). (c, ci) = (block width, inner width cdc = cffn):
// (256, 256), (384, 384), (512, 512), (768, 768), and the half-width `dcb2` blocks (512, 256), (256, 128). Bit-identical to
// the launch sequence. y may alias x (a workgroup reads and writes only its own pixels).
struct DcbNsplitDesc {
    const half_t* t2 = nullptr; int ldt = 0;
    const half_t* x = nullptr; int ldx = 0;
    const half_t* wmain = nullptr;              // packed dc.3 | ffn.0 | ffn.2 (dcb_nsplit_main_halves(c, ci) halves)
    const half_t* wnext = nullptr;              // packed dc.0 of the NEXT block (dcb_nsplit_dc0_halves(c, ci)) or null
    const half_t* b3 = nullptr; const half_t* b0 = nullptr; const half_t* b2 = nullptr; const half_t* b1n = nullptr;
    const half_t* q = nullptr; const half_t* q2 = nullptr;
    half_t* t1n = nullptr; int ldt1 = 0;
    half_t* y = nullptr; int ldy = 0;
    int pixels = 0, c = 0, ci = 0;
    bool shortcut = false;
    // round 6: instead of the next block's dc.0 the launch can run the 1x1 conv that CLOSES a chain (y_prior_fusion.conv.3,
    // y_spatial_prior.conv.3, decoder.conv2, recon_head.head ...): yfin = (Wfin y + bfin) [* qfin] -> fp16, width nfin.
    // wfin = dcb_nsplit_pack_fin's stream. Bit-identical to conv1x1(bias [, q]) on y. With a closing conv y may be null: the
    // block's own output then never leaves LDS (nothing else reads it in the codecs).
    const half_t* wfin = nullptr; const half_t* bfin = nullptr; const half_t* qfin = nullptr;
    half_t* yfin = nullptr; int ldyfin = 0; int nfin = 0;
    // round 6: the block's depthwise conv inside the launch (dcb_nsplit_dw_supported shapes): t1 = dc.0's output [pixels][ldt]
    // INSTEAD of t2, wdw = the taps [9][ci] (tap major, as dwconv3x3 takes them), width = the picture's width (pixels = rows x width).
    // Bit-identical to dwconv3x3 + the launch on its output. t1 must not be the buffer t1n is written to (a workgroup reads its
    // neighbours' rows of t1).
    const half_t* t1 = nullptr; const half_t* wdw = nullptr; int width = 0;
};
bool dcb_nsplit_dw_supported(int c, int ci, int pixels);      // (256, 128); (384, 192) on 32-pixel workgroups. DCVC_NSPLIT_DW=0: never (A/B)
int dcb_nsplit_waves();                                      // 8
bool dcb_nsplit_shape(int c, int ci);                         // a shape the kernel is instantiated for
bool dcb_nsplit_supported(int c, int cdc, int cffn);          // DCVC_NSPLIT: 0 = off, 1 = full-width blocks only (A/B)
size_t dcb_nsplit_main_halves(int c, int ci);
size_t dcb_nsplit_dc0_halves(int c, int ci);
void dcb_nsplit_pack_main(const half_t* w3, const half_t* w0, const half_t* w2, int c, int ci, half_t* out, hipStream_t stream);
void dcb_nsplit_pack_dc0(const half_t* w1, int c, int ci, half_t* out, hipStream_t stream);
bool dcb_nsplit_fin_supported(int c, int ci, int nn);        // a closing conv of width nn behind a (c, ci) block in one launch
size_t dcb_nsplit_fin_halves(int c, int nn);
void dcb_nsplit_pack_fin(const half_t* w /* [nn][c] */, int c, int nn, half_t* out, hipStream_t stream);
void dcb_nsplit(const DcbNsplitDesc& d, hipStream_t stream);
void dcb_nsplit_timeline_buffer(long long* device_buffer);    // tuning aid: [workgroups][32] shader-clock stamps
void dcb_nsplit_dw_hook(const half_t* t1, const half_t* wdw, int width);      // tuning aid: dcb_nsplit launches take their depthwise conv inside, on these operands (null: off)

// The two 1x1 convs in front of a block's depthwise conv in one launch (dcb_pair8_kernel.h, round 6):
//   y = Wa x + ba (the block's adaptor, layers_proxy.cpp:73-77);  t1 = WSiLU(W1 y + b1) (dc.0, :79).
// wa = dcb_pair_pack_adaptor's stream, w1 = the block's dcb_nsplit_pack_dc0 stream. Bit-identical to conv1x1 + conv1x1(wsilu).
struct DcbPairDesc {
    const half_t* x = nullptr; int ldx = 0;
    const half_t* wa = nullptr; const half_t* ba = nullptr;
    const half_t* w1 = nullptr; const half_t* b1 = nullptr;
    half_t* y = nullptr; int ldy = 0;
    half_t* t1 = nullptr; int ldt1 = 0;
    int pixels = 0, cin = 0, c = 0, ci = 0;
};
bool dcb_pair_supported(int cin, int c, int ci);
size_t dcb_pair_adaptor_halves(int cin, int c);
void dcb_pair_pack_adaptor(const half_t* wa /* [c][cin] */, int cin, int c, half_t* out, hipStream_t stream);
void dcb_pair(const DcbPairDesc& d, hipStream_t stream);

struct ConvKxKDesc {
    const half_t* x = nullptr; int ldx = 0;     // [in_h][in_w][ldx]
    const half_t* w = nullptr;                  // [cout][ky][kx][cin]  (tap major, cin contiguous)
    const half_t* bias = nullptr;
    const half_t* zeros = nullptr;              // >= 128 B of zeros
    half_t* y = nullptr; int ldy = 0;           // [out_h][out_w][ldy]
    int in_h = 0, in_w = 0, cin = 0, cout = 0, ksize = 0, stride = 1, pad = 0;
};
void conv_kxk(const ConvKxKDesc& d, hipStream_t stream);

struct TConv2x2Desc {
    const half_t* x = nullptr; int ldx = 0;     // [in_h][in_w][ldx]
    const half_t* w = nullptr;                  // [4 = dy*2+dx][cout][cin]
    half_t* y = nullptr; int ldy = 0;           // [2 in_h][2 in_w][ldy]
    int in_h = 0, in_w = 0, cin = 0, cout = 0;
};
void tconv2x2(const TConv2x2Desc& d, hipStream_t stream);

// ---------------------------------------------------------------- fused FFN (ffn_fused.hip)
// out = W2 * chunk_add(WSiLU(W0 * x + b0)) + b2 + x [+ r2] [* q]  (then [* q2] on the rounded
// output): ffn.0 and ffn.2 of a DepthConvBlock (layers_proxy.cpp:84-98) in one launch, bit-identical
// to conv1x1(wsilu, chunk_add) followed by conv1x1(r1 = x, ...). y may alias x.
struct FfnFusedDesc {
    const half_t* x = nullptr; int ldx = 0;     // [pixels][ldx], first c channels
    const half_t* w0 = nullptr;                 // [4*cffn][c]
    const half_t* b0 = nullptr;                 // [4*cffn]
    const half_t* w2 = nullptr;                 // [c][cffn]
    const half_t* b2 = nullptr;                 // [c]
    const half_t* r2 = nullptr; int ldr2 = 0;   // optional second residual
    const half_t* q = nullptr;                  // optional fused scale
    const half_t* q2 = nullptr;                 // optional scale on the rounded output
    half_t* y = nullptr; int ldy = 0;
    int pixels = 0, c = 0, cffn = 0;
};
// shapes the fused kernel takes (c in {128, 256, 384}, cffn % 64 == 0) AND that are large enough to
// fill the chip with 128-row strips; DCVC_FFN_FUSED=0 switches the fusion off (A/B measurements)
bool ffn_fused_supported(int pixels, int c, int cffn);
void ffn_fused(const FfnFusedDesc& d, hipStream_t stream);

// ---------------------------------------------------------------- DepthConvBlock tail (dcb_tail.hip)
// Everything of a half-width DepthConvBlock behind dc.0 in one launch (layers_proxy.cpp:79-98):
//   t2 = depthwise3x3(t) (when dw != null; else t IS t2), y1 = W3 t2 + b3 + x, then the FFN
//   out = W2 chunk_add(WSiLU(W0 y1 + b0)) + b2 + y1 [+ x when shortcut] [* q], rounded, [* q2].
// c in {128, 256}, cdc <= 128 (half-width blocks, and the full-width 128-wide ones) and cffn multiples of 64. Bit-identical to the
// four-launch sequence.
// y may alias x; t must not alias y.
struct DcbTailDesc {
    const half_t* w1 = nullptr;                 // dc.0 weights [cdc][c] + bias [cdc]: when given (with dw), dc.0 (1x1 +
    const half_t* b1 = nullptr;                 //   WSiLU on x) runs inside the launch as well and `t` is not read
    const half_t* t = nullptr; int ldt = 0;     // dc.0 output [H*W][ldt] (first cdc channels)
    const half_t* dw = nullptr;                 // [9][cdc] tap-major depthwise weights, or null
    const half_t* x = nullptr; int ldx = 0;     // block-internal input (residual of dc.3)
    const half_t* w3 = nullptr;                 // [c][cdc]
    const half_t* b3 = nullptr;                 // [c] (depthwise bias folded in)
    const half_t* w0 = nullptr;                 // [4*cffn][c]
    const half_t* b0 = nullptr;
    const half_t* w2 = nullptr;                 // [c][cffn]
    const half_t* b2 = nullptr;
    const half_t* q = nullptr;
    const half_t* q2 = nullptr;
    half_t* y = nullptr; int ldy = 0;
    int H = 0, W = 0, c = 0, cdc = 0, cffn = 0;
    bool shortcut = false;                      // ffn.2 also adds x (block-level shortcut)
};
// shape supported AND enough 8x16 patches to fill the chip; DCVC_DCB_TAIL=0 / 2 = never / whenever possible
bool dcb_tail_supported(int H, int W, int c, int cdc, int cffn);
bool dcb_tail_takes_dc0();       // false only for A/B measurements (DCVC_DCB_TAIL=3)
void dcb_tail(const DcbTailDesc& d, hipStream_t stream);
// debugging aid: when non-null, launches with dc.0 inside also write dc.0's output of every pixel
// ([H*W][cdc]) to this device buffer
void dcb_tail_debug_buffer(half_t* device_buffer);

// ---------------------------------------------------------------- depthwise 3x3 (dwconv.hip)
// y[h][w][c] = sum_{ky,kx} x[h+ky-1][w+kx-1][c] * wt[ky][kx][c]   (zero padding, no bias: the
// reference folds the bias into the next 1x1, layers_proxy.cpp:175-178)
void dwconv3x3(const half_t* x, int ldx, const half_t* wt /* [9][C] */, half_t* y, int ldy,
               int H, int W, int C, hipStream_t stream);

// ---------------------------------------------------------------- layout kernels (layout.hip)
// x: [H][W][C3] (channels_last view of [1, C3, H, W]); out: [H8][W8][C3*64] with pixel stride
// ldout (0 = dense), replicate padding.
void pad_unshuffle8(const half_t* x, int H, int W, int C3, half_t* out, int H8, int W8,
                    hipStream_t stream, int ldout = 0);
// in: [H8][W8][C3*64] -> out: [H8*8][W8*8][C3] with optional clamp to [-0.5, 0.5]
void shuffle8(const half_t* in, int ldin, int H8, int W8, int C3, bool clamp, half_t* out,
              hipStream_t stream);
// in: [H][W][4*C] -> out [2H][2W][C]   (pixel_shuffle(2), HT-L biased SubpelConv2x)
void shuffle2(const half_t* in, int ldin, int H, int W, int C, half_t* out, int ldout,
              hipStream_t stream);
// replicate pad bottom/right: in [H][W][C] -> out [H+pb][W+pr][C]; with pb = pr = 0 a strided copy
void replicate_pad(const half_t* in, int ldin, int H, int W, int C, int pad_b, int pad_r,
                   half_t* out, int ldout, hipStream_t stream);
// crop: in [Hin][Win][C] -> out [H][W][C]
void crop(const half_t* in, int ldin, int Win, half_t* out, int ldout, int H, int W, int C,
          hipStream_t stream);
// y = x * q[c] (fp16 multiply, may be in place)
void mul_channel(const half_t* x, int ldx, const half_t* q, half_t* y, int ldy, int pixels, int C,
                 hipStream_t stream);

// y = x * max(q, 0.5) (add_and_multiply_with_clamp_min, stream.cu:40-76, after the add) or, with
// `reciprocal`, y = x * fp16(1 / max(q, 0.5)) (divide_with_clamp, stream.cu:422-443); q is a tensor
void scale_clamped(const half_t* x, int ldx, const half_t* q, int ldq, half_t* y, int ldy, int pixels,
                   int C, bool reciprocal, hipStream_t stream);

// ---------------------------------------------------------------- picture I/O (frame_io.hip)
// 8-bit YUV420 planes (y [H][W], uv [2][H/2][W/2]) -> x fp16, 3 channels per pixel at pixel
// stride ldx (3 for one picture, 24 + a channel offset for a chunk of 8): nearest-neighbour
// chroma, x = fp16(fp16(v / 255) - 0.5)   (test_video.py:69-123, transforms.py:69-80)
void yuv420_to_x(const uint8_t* y, const uint8_t* uv, int H, int W, half_t* x, int ldx, hipStream_t stream);
// x_hat fp16 [rows][row_pixels][3] -> the top-left H x W picture as YUV420: fp16 planes scaled to
// 0..255 (what get_distortion measures, test_video.py:32-45) and / or u8 planes (what the writer
// stores, test_video.py:356-363: Y rounded half-to-even, U/V truncated). Null outputs are skipped.
void x_to_yuv420(const half_t* x, int row_pixels, int H, int W, half_t* y16, half_t* uv16, uint8_t* y8,
                 uint8_t* uv8, hipStream_t stream);

// ---------------------------------------------------------------- symbol kernels (symbols.hip)
// Uploads the scale -> Gaussian-table-index lookup table (call once per process before the
// first symbol kernel and outside any graph capture).
void symbols_init();
// number of 2048-element blocks the symbol kernels use for `count` elements (= size of block_count)
int symbol_blocks(int count);

// z -> round half away, clamp [-64, 63] -> fp16 z_hat and int8 symbols
void round_z(const half_t* z, half_t* z_hat, int8_t* z_i8, int count, hipStream_t stream);
void int8_to_half(const int8_t* in, half_t* out, int count, hipStream_t stream);

// One step (0..3) of the 4-step masked quantisation of y (encoder side): fuses the reference's
// process_with_mask + single_part_for_writing_4x (x2) + build_index_enc + compaction counting.
struct YStepEnc {
    const half_t* y = nullptr; int ldy = 0;            // scaled latent [P][C]
    const half_t* scales = nullptr; int lds = 0;       // [P][C]
    const half_t* means = nullptr; int ldm = 0;        // [P][C]
    half_t* y_hat_acc = nullptr; int ldacc = 0;        // y_hat_so_far [P][C] (written for the active group)
    int16_t* sym = nullptr;                            // [P * C/4] (symbol << 8) + index, NHWC order
    uint8_t* cond = nullptr;                           // [P * C/4 / 8] 8 skip flags per byte
    int32_t* block_count = nullptr;                    // [blocks] kept symbols per 2048-element block
    int H = 0, W = 0, C = 0, step = 0;
    float skip_thres = 0.f;
    bool first = false;                                // step 0 initialises y_hat_acc (copy, not add)
};
void y_step_enc(const YStepEnc& d, hipStream_t stream);

// Decoder side: scales of the active group -> table index + skip flag (+ counting)
struct YStepDecIndex {
    const half_t* scales = nullptr; int lds = 0;
    uint8_t* index = nullptr;                          // [P * C/4]
    uint8_t* cond = nullptr;
    int32_t* block_count = nullptr;
    int H = 0, W = 0, C = 0, step = 0;
    float skip_thres = 0.f;
};
void y_step_dec_index(const YStepDecIndex& d, hipStream_t stream);

// Stream compaction out[base + rank] = in[i] for kept i (ELEM = 2: int16 symbols, 1: uint8 index).
// base = total[0..slot) summed; total[slot] receives this step's kept count.
void compact(const void* in, int elem_bytes, const uint8_t* cond, const int32_t* block_count,
             int count, void* out, int32_t* totals, int slot, hipStream_t stream);

// Decoder: scatter decoded int8 symbols back (zeros where skipped), add the mean of the active
// group and accumulate into y_hat_so_far (restore_y_4x*, stream.cu:757-844).
struct YStepDecRestore {
    const int8_t* decoded = nullptr;                   // compacted, this step's symbols start at totals-base
    const uint8_t* cond = nullptr;
    const int32_t* block_count = nullptr;
    const int32_t* totals = nullptr; int slot = 0;
    const half_t* means = nullptr; int ldm = 0;
    half_t* y_hat_acc = nullptr; int ldacc = 0;
    int H = 0, W = 0, C = 0, step = 0;
    bool first = false;
};
void y_step_dec_restore(const YStepDecRestore& d, hipStream_t stream);

// ---------------------------------------------------------------- full-tensor masked steps (inter models)
// The inter models quantise ALL channels of y against one scale tensor in `nsteps` masked steps and
// entropy-code them in one go:
//   nsteps = 2 (LD):   mask_0 = first channel half on even (h + w), second half on odd; mask_1 the
//                      complement (dmc_ld_proxy.cpp:672-683)
//   nsteps = 4 (HT-S): the channel-group x 2x2-position masks of the intra model
//                      (dmc_hts_proxy.cpp:869-890 == common_model.py:174-195)
// A run of 8 channels never straddles a channel group, so every 8-channel vector of a pixel belongs
// to exactly one step.
//
// Encoder step k:
//   k == 0:          y *= 1/max(q_dec, 0.5) in place (divide_with_clamp, stream.cu:422-443) and
//                    y_hat = 0 at the positions of later steps;
//   active positions: quantise against `means` -> symbol (<< 8 | scale index), y_hat = y_q + mean
//                    (process_with_mask_kernel, stream.cu:549-630);
//   k == nsteps - 1: everywhere y_hat *= max(q_dec, 0.5), plus the skip flags / per-block counts of
//                    ALL symbols (build_index_enc, stream.cu:130-161).
struct MaskStepEnc {
    half_t* y = nullptr; int ldy = 0;
    const half_t* q_dec = nullptr; int ldq = 0;
    const half_t* scales = nullptr; int lds = 0;
    const half_t* means = nullptr; int ldm = 0;
    half_t* y_hat = nullptr; int ldh = 0;
    int16_t* sym = nullptr;                 // [P * C] NHWC order
    uint8_t* cond = nullptr;                // [P * C / 8]   (last step)
    int32_t* block_count = nullptr;         // [blocks]      (last step)
    int H = 0, W = 0, C = 0, nsteps = 2, step = 0;
    float skip_thres = 0.f;
};
void mask_step_enc(const MaskStepEnc& d, hipStream_t stream);

// Decoder: table index + skip flag of every symbol (build_index_dec, stream.cu:100-128)
struct MaskDecIndex {
    const half_t* scales = nullptr; int lds = 0;
    uint8_t* index = nullptr;
    uint8_t* cond = nullptr;
    int32_t* block_count = nullptr;
    int H = 0, W = 0, C = 0;
    float skip_thres = 0.f;
};
void mask_dec_index(const MaskDecIndex& d, hipStream_t stream);

// Decoder step k:
//   k == 0:          scatter the decoded symbols back (conditional_recover, stream.cu:360-383); the
//                    symbols of later steps are parked in `yq`, their y_hat is zeroed;
//   active positions: y_hat = y_q + means   (restore_y_kernel, stream.cu:686-729);
//   k == nsteps - 1: everywhere y_hat *= max(q_dec, 0.5).
struct MaskStepDec {
    const int8_t* decoded = nullptr;        // compacted symbols (step 0)
    const uint8_t* cond = nullptr;
    const int32_t* block_count = nullptr;
    const int32_t* totals = nullptr;
    int8_t* yq = nullptr;                   // [P * C] scratch
    const half_t* means = nullptr; int ldm = 0;
    const half_t* q_dec = nullptr; int ldq = 0;
    half_t* y_hat = nullptr; int ldh = 0;
    int H = 0, W = 0, C = 0, nsteps = 2, step = 0;
};
void mask_step_dec(const MaskStepDec& d, hipStream_t stream);

}  // namespace dcvc
