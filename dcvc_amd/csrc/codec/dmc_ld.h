// DMC low-delay inter codec (DCVC-UF "LD") on MI355X. Replaces the reference class DMCLDProxy
// (src/layers/extensions/inference/dmc_ld_proxy.{h,cpp}): set_param / add_ref_feature_from_frame /
// compress / decompress. One picture per call; the temporal state (feature memory, last decoded
// feature, context, temporal prior) stays resident in HBM between calls.
#pragma once

#include "codec/codec_base.h"

namespace dcvc {

class DmcLdCodec : public CodecBase {
public:
    // video_model_ld.py:16-21
    static constexpr int kChSrc = 192, kChY = 128, kChZ = 128, kChD = 256, kChM = 256;

    DmcLdCodec();
    ~DmcLdCodec();

    // dmc_ld_proxy.cpp:595-639
    void set_param(const ParamStore& ps, float skip_thres);

    // dmc_ld_proxy.cpp:407-418. frame: device fp16 [H][W][3], the intra codec's reconstruction
    // (any size; extended by edge replication to multiples of 16 like the pictures themselves).
    void add_ref_feature_from_frame(const half_t* frame, int height, int width, bool apply_adaptor,
                                    hipStream_t stream);

    // dmc_ld_proxy.cpp:420-473. Returns ec_parallel; the bit stream is in stream_bytes().
    int compress(const half_t* x, int height, int width, int qp, bool reset_feature_memory,
                 hipStream_t stream);

    // dmc_ld_proxy.cpp:475-593. x_hat: device fp16 [H16*16][W16*16][3], caller-owned.
    void decompress(const uint8_t* bits, size_t nbytes, int qp, int height, int width, int ec_parallel,
                    bool reset_feature_memory, half_t* x_hat, hipStream_t stream);

    size_t debug_read(const std::string& name, void* dst, size_t cap, hipStream_t stream);

    // Temporal state as one flat DEVICE buffer (reference feature, memory | feature_p, ctx, temporal
    // prior + the validity flags in a 64-byte header): what another GPU needs to continue the same
    // GOP (torch.distributed.send / recv over RCCL, dcvc_amd/sharding.py). export with dst == nullptr
    // returns the size. import requires the same picture size and parameters.
    size_t export_state(void* dst, size_t cap, hipStream_t stream);
    void import_state(const void* src, size_t bytes, int height, int width, hipStream_t stream);

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
    // sub-networks (operands are fixed views of the resident buffers, see prepare())
    // feed_fe: the feature extractor runs RIGHT behind this chain (nothing in between touches the scratch planes): the
    // chain's last launch also computes dc.0 of its first block, and run_fe is told so (dc0_done)
    void run_fa_i(hipStream_t st, bool feed_fe = false);     // FI -> memory
    void run_fa_m(hipStream_t st, bool feed_fe = false);     // [memory | feature_p] -> memory
    void run_fe(hipStream_t st, bool dc0_done = false);      // memory -> ctx
    void run_tpe(hipStream_t st);                     // memory -> temporal params
    void run_encoder(hipStream_t st);                 // [x unshuffled | ctx] -> Y
    void run_hyper_encoder(hipStream_t st);           // Y -> z, z_hat
    void run_priors(hipStream_t st);                  // z_hat, temporal -> common params
    void run_spatial_prior(hipStream_t st);           // [y_hat | common] -> means of step 1
    void run_decoder(hipStream_t st);                 // y_hat, ctx -> feature_p
    void run_recon_head(half_t* x_hat, hipStream_t st);   // feature_p -> FI (+ x_hat)
    void entropy_encode(int qp);                      // worker thread

    // ---- parameters
    DeviceArena m_wmem;
    const half_t *m_q_encoder = nullptr, *m_q_decoder = nullptr, *m_q_feature = nullptr;
    half_t *m_cur_q_encoder = nullptr, *m_cur_q_decoder = nullptr, *m_cur_q_feature = nullptr;
    half_t* m_zeros = nullptr;
    DcbW m_fa_i[4], m_fa_m[4], m_fe[5];
    DcbW m_encb[3];               // encoder.conv1.0, .1 and encoder.conv2: one chain of three blocks, the last with the quant scale
    ConvKW m_enc_down;
    DcbW m_henc0;
    Stride2W m_henc1, m_henc2;
    UpsampleW m_hdec0, m_hdec1;
    DcbW m_hdec2;
    Stride2W m_tpe;
    DcbW m_fus[3];
    FinW m_fus3;
    DcbW m_sp[2];
    FinW m_sp2;
    SubpelW m_dec_up;
    DcbW m_dec1[3];
    FinW m_dec2;
    DcbW m_rh[3];
    FinW m_rh_head;
    float m_skip_thres = 0.f;
    bool m_has_params = false;

    // ---- resident buffers (per resolution)
    Geometry m_g;
    DeviceArena m_bmem;
    Scratch m_s;
    half_t* m_FI = nullptr;      // [P8][192]  reference feature: unshuffled I picture / recon-head output
    half_t* m_CATM = nullptr;    // [P8][512]  memory | feature_p          (feature_adaptor_m input)
    half_t* m_CATD = nullptr;    // [P8][512]  decoder.up out (x unshuffled at 64:256) | ctx
    half_t* m_T = nullptr;       // [P8][256]  chain temporaries (the blocks ping-pong between the two)
    half_t* m_T2 = nullptr;
    half_t *m_Y = nullptr, *m_Ypad = nullptr;
    half_t *m_Z1 = nullptr, *m_Z2 = nullptr, *m_Z3 = nullptr, *m_ZH = nullptr;
    int8_t* m_ZI8 = nullptr;
    half_t *m_H1 = nullptr, *m_H2 = nullptr, *m_HP = nullptr;
    half_t* m_CATPF = nullptr;   // [P16][384] hyper params | temporal params
    half_t* m_CATSP = nullptr;   // [P16][512] y_hat | q_dec | scales | means
    half_t* m_SPT = nullptr;     // [P16][256]
    half_t* m_MEANS1 = nullptr;  // [P16][128]
    int16_t *m_SYM = nullptr, *m_COMP = nullptr;
    uint8_t *m_COND = nullptr, *m_IDX = nullptr, *m_CIDX = nullptr;
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
    bool m_has_ref = false;            // FI valid
    bool m_enc_ready = false;          // memory, ctx and temporal params are those of the next picture to encode
    bool m_memory_has_value = false;   // decoder: the next picture extends the memory (else restarts from FI)
    bool m_has_feature_p = false;
};

}  // namespace dcvc
