// DMC hierarchical inter codecs (DCVC-UF "HT-S" and "HT-L") on MI355X: 8 pictures per call.
// Replaces the reference classes DMCHTSProxy / DMCHTLProxy
// (src/layers/extensions/inference/dmc_hts_proxy.{h,cpp}, dmc_htl_proxy.{h,cpp}).
// One class serves both: the networks differ in depth / block width (read from the checkpoint),
// the entropy stage differs in kind:
//   HT-S  all of y against ONE scale tensor in four masked steps, coded in one go (mask_step_*)
//   HT-L  the intra model's scheme: four symbol groups with their own scales (y_step_*)
#pragma once

#include "codec/codec_base.h"

namespace dcvc {

class DmcHtCodec : public CodecBase {
public:
    // video_model_ht.py:16-23
    static constexpr int kFrames = 8;
    static constexpr int kChSrcI = 192, kChSrc = kChSrcI * kFrames;
    static constexpr int kChY = 256, kChZ = 128, kChD = 512, kChM = 512, kChRecon = 256;

    explicit DmcHtCodec(bool is_hts);
    ~DmcHtCodec();
    bool is_hts() const { return m_hts; }

    // dmc_hts_proxy.cpp:712-760
    void set_param(const ParamStore& ps, float skip_thres);

    // dmc_hts_proxy.cpp:492-502. frame: device fp16 [H][W][3].
    void add_ref_feature_from_frame(const half_t* frame, int height, int width, bool apply_adaptor,
                                    hipStream_t stream);

    // dmc_hts_proxy.cpp:504-585. x: device fp16 [H][W][24] (8 pictures x 3 planes, picture-major
    // channels = the channels_last view of the reference's [1, 24, H, W] input).
    int compress(const half_t* x, int height, int width, int qp, bool reset_feature_memory,
                 hipStream_t stream);

    // dmc_hts_proxy.cpp:587-710. x_hat: device fp16 [8][H16*16][W16*16][3], caller-owned.
    void decompress(const uint8_t* bits, size_t nbytes, int qp, int height, int width, int ec_parallel,
                    bool reset_feature_memory, half_t* x_hat, hipStream_t stream);

    size_t debug_read(const std::string& name, void* dst, size_t cap, hipStream_t stream);

    // Temporal state (reference feature, memory | feature_p, ctx + validity flags) as one flat DEVICE
    // buffer, for moving a running GOP to another GPU (see DmcLdCodec::export_state).
    size_t export_state(void* dst, size_t cap, hipStream_t stream);
    void import_state(const void* src, size_t bytes, int height, int width, hipStream_t stream);

    // Reconstruction-head fan-out (SURVEY 8e iii): the 8 picture heads depend only on feature_p
    // (video_model_ht.py:252-275, dmc_hts_proxy.cpp:325-360), so after decompress() has produced it
    // on the GPU that holds the stream's temporal state, other GPUs can run the heads.
    //   set_recon_mask(m)        decompress() runs only the heads of the pictures in bit mask m
    //                            (bit 7 = the last picture, whose head output is also the reset feature:
    //                            it belongs to the GPU that keeps the state)
    //   export_feature(dst)      feature_p as one dense [P8][512] fp16 device buffer (33.4 MB at 1080p)
    //   import_feature(src,h,w)  the receiving side: installs it into a codec with the same parameters
    //   run_recon_heads(m, out)  heads of mask m -> out + i * picture for every i in m
    void set_recon_mask(unsigned mask);
    size_t export_feature(void* dst, size_t cap, hipStream_t stream);
    void import_feature(const void* src, size_t bytes, int height, int width, hipStream_t stream);
    void run_recon_heads(unsigned mask, half_t* x_hat, hipStream_t stream);

private:
    struct Geometry {
        int H8 = 0, W8 = 0, H16 = 0, W16 = 0, H16p = 0, W16p = 0, H32 = 0, W32 = 0, H64 = 0, W64 = 0;
        int P8() const { return H8 * W8; }
        int P16() const { return H16 * W16; }
        int P16p() const { return H16p * W16p; }
        int P32() const { return H32 * W32; }
        int P64() const { return H64 * W64; }
        bool padded() const { return H16p != H16 || W16p != W16; }
    };

    void prepare(int height, int width);
    void select_qp(int qp, hipStream_t st);
    void run_fa_i(hipStream_t st);                    // FI -> memory
    void run_fa_m(hipStream_t st);                    // [memory | feature_p] -> memory
    void run_fe(hipStream_t st);                      // memory -> ctx
    void run_tpe(hipStream_t st);                     // memory * q_feature -> temporal params
    void run_encoder(hipStream_t st);                 // [x unshuffled | ctx] -> Y
    void run_hyper_encoder(hipStream_t st);           // Y -> z, z_hat
    void run_common(hipStream_t st);                  // z_hat, temporal -> common params
    void run_reduction(hipStream_t st);               // common -> second half of the adaptor input
    void run_spatial_prior(int k, hipStream_t st);    // [y_hat so far | reduced] -> SP
    void run_decoder(hipStream_t st);                 // y_hat, ctx -> feature_p
    void run_recon_head(half_t* x_hat, hipStream_t st);   // feature_p -> 8 pictures + FI
    void run_recon_reset(hipStream_t st);             // feature_p -> FI (picture 7's head only)
    void enc_entropy_stage(hipStream_t st);
    void entropy_encode(int qp);                      // worker thread

    const bool m_hts;
    // ---- parameters
    DeviceArena m_wmem;
    const half_t *m_q_encoder = nullptr, *m_q_decoder = nullptr, *m_q_feature = nullptr;
    half_t *m_cur_q_encoder = nullptr, *m_cur_q_decoder = nullptr, *m_cur_q_feature = nullptr;
    half_t* m_zeros = nullptr;
    DcbChain m_fa_i, m_fa_m, m_fe, m_enc1, m_dec1, m_fus, m_sp;
    ConvKW m_enc_down;
    DcbW m_henc0, m_hdec2;
    Stride2W m_henc1, m_henc2, m_tpe;
    UpsampleW m_hdec0, m_hdec1;
    FinW m_fus3, m_sp3;        // the convs that close the fusion / spatial prior chains
    Conv1x1W m_reduction;
    DcbW m_sp_adaptor[3];
    SubpelW m_dec_up;
    DcbW m_rh_common[kFrames / 2];          // HT-S: recon_head.conv1.i.0
    DcbChain m_rh[kFrames];                 // HT-S: recon_head.conv2.i.{0,1,2}; HT-L: recon_head.conv.i.{0..4}
    FinW m_rh_head[kFrames];
    float m_skip_thres = 0.f;
    bool m_has_params = false;

    // ---- resident buffers (per resolution)
    Geometry m_g;
    DeviceArena m_bmem;
    Scratch m_s;
    half_t* m_FI = nullptr;      // [P8][192]   reference feature
    half_t* m_CATM = nullptr;    // [P8][1024]  memory | feature_p
    half_t* m_CATE = nullptr;    // [P8][2048]  x unshuffled (1536) | ctx (512); decoder.up out at 1024:1536
    half_t* m_T = nullptr;       // [P8][512]   chain temporary
    half_t* m_TI = nullptr;      // [P8][512]   memory * q_feature
    half_t* m_RC = nullptr;      // [P8][512]   recon head: shared trunk of a picture pair (HT-S)
    half_t* m_RT = nullptr;      // [P8][256]   recon head: per-picture chain
    half_t* m_RH = nullptr;      // [P8][192]   recon head: head output of pictures 0..6
    half_t* m_UPT = nullptr;     // biased upsampler temporary (HT-L)
    half_t *m_Y = nullptr, *m_Ypad = nullptr;
    half_t *m_Z1 = nullptr, *m_Z2 = nullptr, *m_Z3 = nullptr, *m_ZH = nullptr;
    int8_t* m_ZI8 = nullptr;
    half_t *m_H1 = nullptr, *m_H2 = nullptr, *m_HP = nullptr;
    half_t* m_CATPF = nullptr;   // [P16][768]  hyper params | temporal params
    half_t* m_COMMON = nullptr;  // [P16][768]  q_dec | scales | means
    half_t* m_CATSP = nullptr;   // [P16][512]  y_hat so far | reduced params
    half_t* m_AD = nullptr;      // [P16][512]
    half_t* m_SP = nullptr;      // [P16][256] means (HT-S) / [P16][512] scales | means (HT-L)
    int16_t *m_SYM = nullptr, *m_COMP = nullptr;
    uint8_t *m_COND = nullptr, *m_IDX = nullptr, *m_CIDX = nullptr;
    size_t m_idx_region = 0;          // bytes per decode round trip in m_CIDX / m_h_idx: 16 (count) + its symbols
    static constexpr size_t kFirstIdxCopy = 192 * 1024;      // index bytes the first copy of a round trip takes along
    int8_t *m_DECODED = nullptr, *m_YQ = nullptr;
    int32_t *m_CNT = nullptr, *m_TOTALS = nullptr;
    Pinned<int32_t> m_h_totals;
    Pinned<int16_t> m_h_sym;
    Pinned<int8_t> m_h_z;
    Pinned<uint8_t> m_h_idx;
    Pinned<int8_t> m_h_dec;
    hipEvent_t m_ev_idx = nullptr;
    int m_ec_parallel = 1;

    // ---- temporal state flags
    bool m_has_ref = false;
    bool m_enc_ready = false;          // memory and ctx are those of the next chunk to encode
    bool m_memory_has_value = false;
    bool m_has_feature_p = false;
    unsigned m_recon_mask = 0xffu;     // pictures whose heads decompress() runs
};

}  // namespace dcvc
