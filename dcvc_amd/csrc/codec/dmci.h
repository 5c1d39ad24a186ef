// DMCI (DCVC-UF intra picture codec) on MI355X. Replaces the reference class DMCIProxy
// (src/layers/extensions/inference/dmci_proxy.{h,cpp}): set_param / compress / decompress.
#pragma once

#include "codec/codec_base.h"

namespace dcvc {

class DmciCodec : public CodecBase {
public:
    static constexpr int kChSrc = 192, kChEncDec = 384, kChY = 256, kChZ = 128;   // image_model.py:15-18

    DmciCodec();
    ~DmciCodec();

    // dmci_proxy.cpp:604-652. All tensors in host memory (see ParamStore::add).
    void set_param(const ParamStore& ps, float skip_thres);

    // dmci_proxy.cpp:296-421. x: device fp16 [H][W][3] (unpadded, values in [-0.5, 0.5]);
    // x_hat: device fp16 [H16][W16][3] (caller-owned). The bit stream is available through
    // stream_bytes() when the call returns; the reconstruction kernels may still be running on
    // `stream` (as in the reference, the harness synchronises).
    int compress(const half_t* x, int height, int width, int qp, half_t* x_hat, hipStream_t stream);

    // dmci_proxy.cpp:423-602
    void decompress(const uint8_t* bits, size_t nbytes, int qp, int height, int width, int ec_parallel,
                    half_t* x_hat, hipStream_t stream);

    // test hook: copies an internal tensor of the last call to the host. Returns the byte size.
    size_t debug_read(const std::string& name, void* dst, size_t cap, hipStream_t stream);

private:
    struct Geometry {
        int H = 0, W = 0;           // picture
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
    // network stages
    void run_encoder(hipStream_t st);                       // U -> Y
    void run_hyper_and_priors_enc(hipStream_t st);          // Y -> z, params, reduced
    void run_priors_from_zhat(hipStream_t st);              // ZH -> params, reduced
    void run_spatial_prior(int k, hipStream_t st);          // CAT -> SP (scales | means)
    void run_decoder(half_t* x_hat, hipStream_t st);        // YHAT -> x_hat
    void enc_stage0(hipStream_t st);
    void entropy_encode(int qp);                            // worker thread

    // ---- parameters
    DeviceArena m_wmem;
    const half_t* m_q_enc = nullptr;      // [64][384]
    const half_t* m_q_dec = nullptr;
    const half_t* m_q_y_enc = nullptr;    // [64][256]
    const half_t* m_q_y_dec = nullptr;
    half_t* m_zeros = nullptr;
    DcbW m_enc1, m_enc2[6];
    ConvKW m_enc_down;
    DcbW m_henc0;
    Stride2W m_henc1, m_henc2;
    UpsampleW m_hdec0, m_hdec1;
    DcbW m_hdec2;
    DcbW m_fus[3];
    FinW m_fus3;               // y_prior_fusion.conv.3: closes the fusion chain (inside its last block launch)
    Conv1x1W m_reduction;
    DcbW m_sp_adaptor[3], m_sp[3];
    FinW m_sp3;                // y_spatial_prior.conv.3: closes the spatial prior chain
    UpsampleW m_dec_up;
    DcbW m_dec1[12], m_dec2;
    float m_skip_thres = 0.f;
    bool m_has_params = false;

    // ---- per-resolution buffers
    Geometry m_g;
    DeviceArena m_bmem;
    Scratch m_s;
    half_t *m_U = nullptr, *m_F = nullptr, *m_Y = nullptr, *m_Ypad = nullptr;
    half_t *m_Z1 = nullptr, *m_Z2a = nullptr, *m_Z2 = nullptr, *m_Z3a = nullptr, *m_Z3 = nullptr, *m_ZH = nullptr;
    int8_t* m_ZI8 = nullptr;
    half_t *m_H1a = nullptr, *m_H1 = nullptr, *m_H2a = nullptr, *m_H2 = nullptr, *m_HP = nullptr;
    half_t *m_PF = nullptr, *m_PARAMSp = nullptr, *m_PARAMS = nullptr, *m_CAT = nullptr, *m_AD = nullptr, *m_SP = nullptr;
    half_t *m_YHAT = nullptr, *m_D0 = nullptr, *m_D1 = nullptr, *m_R = nullptr;
    half_t *m_cur_q_enc = nullptr, *m_cur_q_dec = nullptr, *m_cur_q_y_enc = nullptr, *m_cur_q_y_dec = nullptr;
    int16_t *m_SYM = nullptr, *m_COMP = nullptr;
    uint8_t *m_COND = nullptr, *m_IDX = nullptr, *m_CIDX = nullptr;
    int8_t* m_DECODED = nullptr;
    size_t m_idx_region = 0;           // bytes per decode step in m_CIDX / m_h_idx: 16 (count) + symbols of a step
    int32_t *m_CNT = nullptr, *m_TOTALS = nullptr;
    // pinned host staging
    Pinned<int32_t> m_h_totals;
    Pinned<int16_t> m_h_sym;
    Pinned<int8_t> m_h_z;
    Pinned<uint8_t> m_h_idx;
    Pinned<int8_t> m_h_dec;
    int m_ec_parallel = 1;
};

}  // namespace dcvc
