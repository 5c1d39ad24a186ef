// DMC HT-S / HT-L codec orchestration on MI355X (see dmc_ht.h). Dataflow follows
// dmc_hts_proxy.cpp:492-710 and dmc_htl_proxy.cpp:583-905; launch structure as in the other codecs
// (one hipGraph per stage for all qps, concatenations = channel ranges of resident buffers).
#include "codec/dmc_ht.h"

#include <algorithm>

namespace dcvc {

namespace {

enum StageKey : int { kRef = 0, kEnc0 = 1, kEnc1 = 2 /* +reset */, kDec0 = 10 /* +memory_has_value */,
                      kDec1 = 12, kDec2 = 13, kDecStep = 20 /* +k */ };

}  // namespace

DmcHtCodec::DmcHtCodec(bool is_hts) : m_hts(is_hts)
{
    hip_check(hipEventCreateWithFlags(&m_ev_idx, hipEventDisableTiming), "hipEventCreate");
}

DmcHtCodec::~DmcHtCodec()
{
    quiesce();
    if (m_ev_idx) (void)hipEventDestroy(m_ev_idx);
}

// ------------------------------------------------------------------------------------ set_param
void DmcHtCodec::set_param(const ParamStore& ps, float skip_thres)
{
    quiesce();
    clear_graphs();
    m_wmem.release();
    kernels_init();
    m_skip_thres = skip_thres;
    const bool want_hts = ps.has("recon_head.conv1.0.0.dc.0.weight");
    if (want_hts != m_hts) {
        throw std::invalid_argument(m_hts ? "DMCHTSProxy was given an HT-L state_dict"
                                          : "DMCHTLProxy was given an HT-S state_dict");
    }
    m_q_encoder = upload_qp_table(ps, m_wmem, "q_encoder", kChD);
    m_q_decoder = upload_qp_table(ps, m_wmem, "q_decoder", kChD);
    m_q_feature = upload_qp_table(ps, m_wmem, "q_feature", kChD);
    m_cur_q_encoder = m_wmem.alloc_half(kChD);
    m_cur_q_decoder = m_wmem.alloc_half(kChD);
    m_cur_q_feature = m_wmem.alloc_half(kChD);
    m_zeros = m_wmem.alloc_half(2048);
    const bool sc = !m_hts;          // block-level shortcuts of the hyper / temporal nets (HT-L only)
    m_fa_i.load(ps, m_wmem, "feature_adaptor_i.conv.");
    m_fa_m.load(ps, m_wmem, "feature_adaptor_m.conv.");
    m_fe.load(ps, m_wmem, "feature_extractor.conv.");
    m_enc1.load(ps, m_wmem, "encoder.conv1.");
    m_enc_down.load(ps, m_wmem, "encoder.down.");
    m_henc0.load(ps, m_wmem, "hyper_encoder.conv.0.");
    m_henc1.load(ps, m_wmem, "hyper_encoder.conv.1.", sc);
    m_henc2.load(ps, m_wmem, "hyper_encoder.conv.2.", sc);
    m_hdec0.load(ps, m_wmem, "hyper_decoder.conv.0.", sc);
    m_hdec1.load(ps, m_wmem, "hyper_decoder.conv.1.", sc);
    m_hdec2.load(ps, m_wmem, "hyper_decoder.conv.2.");
    m_tpe.load(ps, m_wmem, "temporal_prior_encoder.conv.", sc);
    m_fus.load(ps, m_wmem, "y_prior_fusion.conv.");
    m_fus3.load(ps, m_wmem, "y_prior_fusion.conv.3.");
    m_reduction.load(ps, m_wmem, "y_spatial_prior_reduction.");
    for (int i = 0; i < 3; ++i) {
        m_sp_adaptor[i].load(ps, m_wmem, "y_spatial_prior_adaptor_" + std::to_string(i + 1) + ".");
    }
    m_sp.load(ps, m_wmem, "y_spatial_prior.conv.");
    m_sp3.load(ps, m_wmem, "y_spatial_prior.conv.3.");
    m_dec_up.load(ps, m_wmem, "decoder.up.");
    m_dec1.load(ps, m_wmem, "decoder.conv1.");
    for (int i = 0; i < kFrames; ++i) {
        const std::string n = std::to_string(i);
        if (m_hts) {
            if (i % 2 == 0) m_rh_common[i / 2].load(ps, m_wmem, "recon_head.conv1." + std::to_string(i / 2) + ".0.");
            m_rh[i].load(ps, m_wmem, "recon_head.conv2." + n + ".");
            m_rh_head[i].load(ps, m_wmem, "recon_head.conv2." + n + "." + std::to_string(m_rh[i].size()) + ".");
        } else {
            m_rh[i].load(ps, m_wmem, "recon_head.conv." + n + ".");
            m_rh_head[i].load(ps, m_wmem, "recon_head.conv." + n + "." + std::to_string(m_rh[i].size()) + ".");
        }
    }
    if (m_sp3.conv.cout != (m_hts ? kChY : 2 * kChY)) throw std::invalid_argument("unexpected y_spatial_prior.conv.3 width");
    load_cdf_tables(ps);
    m_has_params = true;
    m_has_ref = m_enc_ready = m_memory_has_value = m_has_feature_p = false;
}

// ------------------------------------------------------------------------------------ buffers
void DmcHtCodec::prepare(int height, int width)
{
    if (!m_has_params) throw std::runtime_error("DMC-HT: set_param() has not been called");
    if (height <= 0 || width <= 0) throw std::invalid_argument("DMC-HT: empty picture");
    const int H8 = ceil_div(height, 16) * 2, W8 = ceil_div(width, 16) * 2;
    if (m_g.H8 == H8 && m_g.W8 == W8) return;
    quiesce();
    clear_graphs();
    m_bmem.release();
    m_has_ref = m_enc_ready = m_memory_has_value = m_has_feature_p = false;
    Geometry g;
    g.H8 = H8; g.W8 = W8;
    g.H16 = H8 / 2; g.W16 = W8 / 2;
    g.H16p = ceil_div(g.H16, 4) * 4; g.W16p = ceil_div(g.W16, 4) * 4;      // dmc_common.cpp:73-83
    g.H32 = g.H16p / 2; g.W32 = g.W16p / 2;
    g.H64 = g.H16p / 4; g.W64 = g.W16p / 4;
    m_g = g;
    auto H = [&](size_t n) { return m_bmem.alloc_half(n); };
    const size_t P8 = g.P8(), P16 = g.P16(), P16p = g.P16p(), P32 = g.P32(), P64 = g.P64();
    m_s.elems = std::max<size_t>(P8 * kChD, P16p * 3 * kChY);
    m_s.t1 = H(m_s.elems); m_s.t2 = H(m_s.elems); m_s.t3 = H(m_s.elems);
    m_FI = H(P8 * kChSrcI);
    m_CATM = H(P8 * (kChM + kChD));
    m_CATE = H(P8 * (kChSrc + kChD));
    m_T = H(P8 * kChD); m_TI = H(P8 * kChM);
    m_RC = H(P8 * kChD); m_RT = H(P8 * kChRecon); m_RH = H(P8 * kChSrcI);
    const size_t upt = std::max({ m_hdec0.up.tmp_elems(g.H64, g.W64), m_hdec1.up.tmp_elems(g.H32, g.W32),
                                  m_dec_up.tmp_elems(g.H16, g.W16) });
    m_UPT = upt ? H(upt) : nullptr;
    m_Y = H(P16 * kChY); m_Ypad = g.padded() ? H(P16p * kChY) : m_Y;
    m_Z1 = H(P16p * kChY); m_Z2 = H(2 * P32 * kChY); m_Z3 = H(2 * P64 * kChZ); m_ZH = H(P64 * kChZ);
    m_ZI8 = static_cast<int8_t*>(m_bmem.alloc(P64 * kChZ));
    m_H1 = H(2 * P32 * kChY); m_H2 = H(2 * P16p * kChY); m_HP = H(P16p * kChY);
    m_CATPF = H(P16 * 3 * kChY);
    m_COMMON = H(P16 * 3 * kChY);
    m_CATSP = H(P16 * 2 * kChY);
    m_AD = H(P16 * 2 * kChY);
    m_SP = H(P16 * 2 * kChY);
    const size_t n = P16 * kChY;                 // all symbols of a chunk
    m_SYM = static_cast<int16_t*>(m_bmem.alloc(n * 2));
    m_COMP = static_cast<int16_t*>(m_bmem.alloc(n * 2));
    m_COND = static_cast<uint8_t*>(m_bmem.alloc(n / 8 + 8));
    m_IDX = static_cast<uint8_t*>(m_bmem.alloc(n));
    // decode side: [count, int32 | 12 B pad | compacted indexes] per round trip (HT-S: one of n symbols, HT-L: four
    // of n / 4), so that ONE device->host copy brings the count and (nearly always) all the indexes
    m_idx_region = (16 + (m_hts ? n : n / 4) + 15) / 16 * 16;
    m_CIDX = static_cast<uint8_t*>(m_bmem.alloc((m_hts ? 1 : 4) * m_idx_region));
    m_DECODED = static_cast<int8_t*>(m_bmem.alloc(n));
    m_YQ = static_cast<int8_t*>(m_bmem.alloc(n));
    m_CNT = static_cast<int32_t*>(m_bmem.alloc(sizeof(int32_t) * symbol_blocks(static_cast<int>(n))));
    m_TOTALS = static_cast<int32_t*>(m_bmem.alloc(sizeof(int32_t) * 4));
    m_h_totals.reserve(16);
    m_h_sym.reserve(n);
    m_h_z.reserve(P64 * kChZ + 64);
    m_h_idx.reserve((m_hts ? 1 : 4) * m_idx_region);
    m_h_dec.reserve(n);
}

void DmcHtCodec::select_qp(int qp, hipStream_t st)
{
    copy_qp_rows({{m_cur_q_encoder, m_q_encoder, kChD}, {m_cur_q_decoder, m_q_decoder, kChD},
                  {m_cur_q_feature, m_q_feature, kChD}}, qp, st);
}

// ------------------------------------------------------------------------------------ networks
void DmcHtCodec::run_fa_i(hipStream_t st)
{
    const View t(m_T, kChM, kChM);
    m_fa_i.forward(View(m_FI, kChSrcI, kChSrcI), t, View(m_CATM, kChM + kChD, kChM), m_g.H8, m_g.W8, m_s, st);
}

void DmcHtCodec::run_fa_m(hipStream_t st)
{
    const View t(m_T, kChM, kChM);
    m_fa_m.forward(View(m_CATM, kChM + kChD, kChM + kChD), t, View(m_CATM, kChM + kChD, kChM),
                   m_g.H8, m_g.W8, m_s, st);
}

void DmcHtCodec::run_fe(hipStream_t st)
{
    const View t(m_T, kChD, kChD);
    m_fe.forward(View(m_CATM, kChM + kChD, kChM), t, View(m_CATE + kChSrc, kChSrc + kChD, kChD),
                 m_g.H8, m_g.W8, m_s, st);
}

void DmcHtCodec::run_tpe(hipStream_t st)
{
    const Geometry& g = m_g;
    // multiply_with_broadcast(memory, q_feature) -> temporal input (dmc_hts_proxy.cpp:525)
    mul_channel(m_CATM, kChM + kChD, m_cur_q_feature, m_TI, kChM, g.P8(), kChM, st);
    const View out(m_CATPF + kChY, 3 * kChY, 2 * kChY);
    const View tmp = m_tpe.shortcut ? View(m_AD, 2 * kChY, 2 * kChY) : out;     // AD is free at this point
    m_tpe.forward(View(m_TI, kChM, kChM), tmp, out, g.H8, g.W8, m_zeros, m_s, st);
}

void DmcHtCodec::run_encoder(hipStream_t st)
{
    const Geometry& g = m_g;
    const View t(m_T, kChD, kChD);
    m_enc1.forward(View(m_CATE, kChSrc + kChD, kChSrc + kChD), t, t, g.H8, g.W8, m_s, st, m_cur_q_encoder);
    ConvKxKDesc d;
    d.x = m_T; d.ldx = kChD; d.w = m_enc_down.w; d.bias = m_enc_down.b; d.zeros = m_zeros;
    d.y = m_Y; d.ldy = kChY; d.in_h = g.H8; d.in_w = g.W8; d.cin = kChD; d.cout = kChY;
    d.ksize = 3; d.stride = 2; d.pad = 1;
    conv_kxk(d, st);
}

void DmcHtCodec::run_hyper_encoder(hipStream_t st)
{
    const Geometry& g = m_g;
    if (g.padded()) {
        replicate_pad(m_Y, kChY, g.H16, g.W16, kChY, g.H16p - g.H16, g.W16p - g.W16, m_Ypad, kChY, st);
    }
    const View z1(m_Z1, kChY, kChY);
    const View z2a(m_Z2, kChY, kChY), z2(m_Z2 + static_cast<size_t>(g.P32()) * kChY, kChY, kChY);
    const View z3a(m_Z3, kChZ, kChZ), z3(m_Z3 + static_cast<size_t>(g.P64()) * kChZ, kChZ, kChZ);
    m_henc0.forward(View(m_Ypad, kChY, kChY), z1, g.H16p, g.W16p, m_s, st);
    m_henc1.forward(z1, z2a, z2, g.H16p, g.W16p, m_zeros, m_s, st);
    m_henc2.forward(z2, z3a, z3, g.H32, g.W32, m_zeros, m_s, st);
    round_z(z3.p, m_ZH, m_ZI8, g.P64() * kChZ, st);
}

void DmcHtCodec::run_common(hipStream_t st)
{
    const Geometry& g = m_g;
    const View h1a(m_H1, kChY, kChY), h1(m_H1 + static_cast<size_t>(g.P32()) * kChY, kChY, kChY);
    const View h2a(m_H2, kChY, kChY), h2(m_H2 + static_cast<size_t>(g.P16p()) * kChY, kChY, kChY);
    m_hdec0.forward(View(m_ZH, kChZ, kChZ), h1a, h1, g.H64, g.W64, m_s, st, m_UPT, m_zeros);
    m_hdec1.forward(h1, h2a, h2, g.H32, g.W32, m_s, st, m_UPT, m_zeros);
    m_hdec2.forward(h2, View(m_HP, kChY, kChY), g.H16p, g.W16p, m_s, st);
    crop(m_HP, kChY, g.W16p, m_CATPF, 3 * kChY, g.H16, g.W16, kChY, st);       // crop_hyper_params
    const View pf(m_CATPF, 3 * kChY, 3 * kChY);
    const FinCall fin(m_fus3, m_COMMON, 3 * kChY);          // y_prior_fusion.conv.3 closes the chain
    m_fus.forward(pf, pf, pf, g.H16, g.W16, m_s, st, nullptr, View(), &fin);
}

void DmcHtCodec::run_reduction(hipStream_t st)
{
    Conv1x1Desc d;
    d.x = m_COMMON; d.ldx = 3 * kChY; d.w = m_reduction.w; d.bias = m_reduction.b;
    d.y = m_CATSP + kChY; d.ldy = 2 * kChY; d.pixels = m_g.P16(); d.cin = 3 * kChY; d.cout = kChY;
    conv1x1(d, st);
}

void DmcHtCodec::run_spatial_prior(int k, hipStream_t st)
{
    const Geometry& g = m_g;
    const View ad(m_AD, 2 * kChY, 2 * kChY);
    m_sp_adaptor[k].forward(View(m_CATSP, 2 * kChY, 2 * kChY), ad, g.H16, g.W16, m_s, st);
    const FinCall fin(m_sp3, m_SP, m_sp3.conv.cout);       // y_spatial_prior.conv.3 closes the chain
    m_sp.forward(ad, ad, ad, g.H16, g.W16, m_s, st, nullptr, View(), &fin);
}

void DmcHtCodec::run_decoder(hipStream_t st)
{
    const Geometry& g = m_g;
    const int ld = kChSrc + kChD;
    half_t* cat = m_CATE + (kChSrc - kChD);                 // [up out | ctx], dmc_hts_proxy.cpp:936-941
    m_dec_up.forward(View(m_CATSP, 2 * kChY, kChY), View(cat, ld, kChD), g.H16, g.W16, st, m_UPT, m_zeros);
    const View t(m_T, kChD, kChD);
    m_dec1.forward(View(cat, ld, 2 * kChD), t, View(m_CATM + kChM, kChM + kChD, kChD), g.H8, g.W8, m_s, st,
                   m_cur_q_decoder);
}

void DmcHtCodec::run_recon_head(half_t* x_hat, hipStream_t st)
{
    const Geometry& g = m_g;
    const View feature(m_CATM + kChM, kChM + kChD, kChD);
    const View rc(m_RC, kChD, kChD), rt(m_RT, kChRecon, kChRecon);
    const size_t picture = static_cast<size_t>(g.P8()) * 64 * 3;
    bool trunk_done = false;       // HT-S: pictures 2j and 2j+1 share the trunk recon_head.conv1.j
    for (int i = 0; i < kFrames; ++i) {
        if (i % 2 == 0) trunk_done = false;
        if (!(m_recon_mask >> i & 1u)) continue;
        View trunk = feature;
        if (m_hts) {
            if (!trunk_done) m_rh_common[i / 2].forward(feature, rc, g.H8, g.W8, m_s, st);
            trunk_done = true;
            trunk = rc;
        }
        half_t* head = (i == kFrames - 1) ? m_FI : m_RH;    // the last head output doubles as the reset feature
        const FinCall fin(m_rh_head[i], head, kChSrcI);     // the head conv closes the chain
        m_rh[i].forward(trunk, rt, rt, g.H8, g.W8, m_s, st, nullptr, View(), &fin);
        shuffle8(head, kChSrcI, g.H8, g.W8, 3, true, x_hat + i * picture, st);
    }
}

void DmcHtCodec::run_recon_reset(hipStream_t st)
{
    // forward_reset, dmc_hts_proxy.cpp:346-358
    const Geometry& g = m_g;
    const View feature(m_CATM + kChM, kChM + kChD, kChD);
    const View rc(m_RC, kChD, kChD), rt(m_RT, kChRecon, kChRecon);
    const int i = kFrames - 1;
    View trunk = feature;
    if (m_hts) {
        m_rh_common[i / 2].forward(feature, rc, g.H8, g.W8, m_s, st);
        trunk = rc;
    }
    const FinCall fin(m_rh_head[i], m_FI, kChSrcI);
    m_rh[i].forward(trunk, rt, rt, g.H8, g.W8, m_s, st, nullptr, View(), &fin);
}

// ------------------------------------------------------------------------------------ reference frame
void DmcHtCodec::add_ref_feature_from_frame(const half_t* frame, int height, int width,
                                            bool apply_adaptor, hipStream_t user)
{
    prepare(height, width);
    hipStream_t st = enter(user);
    pad_unshuffle8(frame, height, width, 3, m_FI, m_g.H8, m_g.W8, st);
    if (apply_adaptor) {
        run_stage(kRef, st, [&] {
            run_fa_i(st);
            run_fe(st);
        });
    }
    leave(user);
    m_has_ref = true;
    m_enc_ready = apply_adaptor;
    m_memory_has_value = apply_adaptor;
    m_has_feature_p = false;
}

// ------------------------------------------------------------------------------------ compress
void DmcHtCodec::enc_entropy_stage(hipStream_t st)
{
    const Geometry& g = m_g;
    const int ldc = 3 * kChY;
    if (m_hts) {
        MaskStepEnc d;
        d.y = m_Y; d.ldy = kChY;
        d.q_dec = m_COMMON; d.ldq = ldc;
        d.scales = m_COMMON + kChY; d.lds = ldc;
        d.y_hat = m_CATSP; d.ldh = 2 * kChY;
        d.sym = m_SYM; d.cond = m_COND; d.block_count = m_CNT;
        d.H = g.H16; d.W = g.W16; d.C = kChY; d.nsteps = 4; d.skip_thres = m_skip_thres;
        for (int k = 0; k < 4; ++k) {
            if (k == 0) { d.means = m_COMMON + 2 * kChY; d.ldm = ldc; }
            else { run_spatial_prior(k - 1, st); d.means = m_SP; d.ldm = kChY; }
            d.step = k;
            mask_step_enc(d, st);
        }
        compact(m_SYM, 2, m_COND, m_CNT, g.P16() * kChY, m_COMP, m_TOTALS, 0, st);
        return;
    }
    // HT-L: divide, then the intra model's four-group scheme (dmc_htl_proxy.cpp:624-689)
    scale_clamped(m_Y, kChY, m_COMMON, ldc, m_Y, kChY, g.P16(), kChY, true, st);
    const int nq = g.P16() * (kChY / 4);
    for (int k = 0; k < 4; ++k) {
        YStepEnc d;
        d.y = m_Y; d.ldy = kChY;
        if (k == 0) { d.scales = m_COMMON + kChY; d.lds = ldc; d.means = m_COMMON + 2 * kChY; d.ldm = ldc; }
        else { d.scales = m_SP; d.lds = 2 * kChY; d.means = m_SP + kChY; d.ldm = 2 * kChY; }
        d.y_hat_acc = m_CATSP; d.ldacc = 2 * kChY;
        d.sym = m_SYM; d.cond = m_COND; d.block_count = m_CNT;
        d.H = g.H16; d.W = g.W16; d.C = kChY; d.step = k; d.skip_thres = m_skip_thres; d.first = (k == 0);
        y_step_enc(d, st);
        compact(m_SYM, 2, m_COND, m_CNT, nq, m_COMP, m_TOTALS, k, st);
        if (k < 3) run_spatial_prior(k, st);
    }
    // add_and_multiply_with_clamp_min: (y_hat_3 + y_hat_so_far) * max(q_dec, 0.5)
    scale_clamped(m_CATSP, 2 * kChY, m_COMMON, ldc, m_CATSP, 2 * kChY, g.P16(), kChY, false, st);
}

int DmcHtCodec::compress(const half_t* x, int height, int width, int qp, bool reset, hipStream_t user)
{
    prepare(height, width);
    if (!m_enc_ready) {
        throw std::runtime_error("DMC-HT compress: no reference feature "
                                 "(call add_ref_feature_from_frame(frame, true) first)");
    }
    const Geometry& g = m_g;
    hipStream_t st = enter(user);
    select_qp(qp, st);
    pad_unshuffle8(x, height, width, 3 * kFrames, m_CATE, g.H8, g.W8, st, kChSrc + kChD);
    run_stage(kEnc0, st, [&] {
        run_encoder(st);
        run_hyper_encoder(st);
        run_tpe(st);
        run_common(st);
        run_reduction(st);
        enc_entropy_stage(st);
    });
    submit(st, [this, qp] { entropy_encode(qp); });
    run_stage(kEnc1 + (reset ? 1 : 0), st, [&] {
        run_decoder(st);
        if (reset) {
            run_recon_reset(st);
            run_fa_i(st);
        } else {
            run_fa_m(st);
        }
        run_fe(st);
    });
    leave(user);
    m_has_feature_p = true;
    wait_job();
    return m_ec_parallel;
}

void DmcHtCodec::entropy_encode(int qp)
{
    const Geometry& g = m_g;
    const int groups = m_hts ? 1 : 4;
    hip_check(hipMemcpyAsync(m_h_totals.get(), m_TOTALS, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, m_io_stream), "D2H totals");
    const int nz = g.P64() * kChZ;
    hip_check(hipMemcpyAsync(m_h_z.get(), m_ZI8, nz, hipMemcpyDeviceToHost, m_io_stream), "D2H z");
    hip_check(hipStreamSynchronize(m_io_stream), "sync io");
    int base[4] = { 0, 0, 0, 0 }, total = 0;
    for (int k = 0; k < groups; ++k) {
        base[k] = total;
        total += m_h_totals[k];
    }
    if (total > 0) {
        hip_check(hipMemcpyAsync(m_h_sym.get(), m_COMP, static_cast<size_t>(total) * 2, hipMemcpyDeviceToHost, m_io_stream), "D2H symbols");
        hip_check(hipStreamSynchronize(m_io_stream), "sync io");
    }
    m_ec_parallel = ec_parallel_for(total);
    m_enc.reset();
    m_enc.set_parallel(m_ec_parallel);
    for (int k = groups - 1; k >= 0; --k) m_enc.push_y(m_h_sym.get() + base[k], m_h_totals[k]);
    m_enc.push_z(m_h_z.get(), nz, qp * kChZ, kChZ);
    m_enc.flush();
}

// ------------------------------------------------------------------------------------ decompress
void DmcHtCodec::decompress(const uint8_t* bits, size_t nbytes, int qp, int height, int width,
                            int ec_parallel, bool reset, half_t* x_hat, hipStream_t user)
{
    prepare(height, width);
    if (m_memory_has_value ? !m_has_feature_p : !m_has_ref) {
        throw std::runtime_error("DMC-HT decompress: no reference feature "
                                 "(call add_ref_feature_from_frame first)");
    }
    const Geometry& g = m_g;
    const int ldc = 3 * kChY;
    hipStream_t st = enter(user);
    select_qp(qp, st);
    const bool extend = m_memory_has_value;
    run_stage(kDec0 + (extend ? 1 : 0), st, [&] {
        if (extend) run_fa_m(st);
        else run_fa_i(st);
        run_tpe(st);
    });
    m_dec.set_parallel(ec_parallel);
    m_dec.set_stream(bits, nbytes);
    const int nz = g.P64() * kChZ;
    m_dec.decode_z(nz, qp * kChZ, kChZ, m_h_z.get());
    hip_check(hipMemcpyAsync(m_ZI8, m_h_z.get(), nz, hipMemcpyHostToDevice, st), "H2D z");
    bind_stage_arg(kDecStep + 3, x_hat);

    if (m_hts) {
        const int ny = g.P16() * kChY;
        run_stage(kDec1, st, [&] {
            int8_to_half(m_ZI8, m_ZH, nz, st);
            run_common(st);
            MaskDecIndex d;
            d.scales = m_COMMON + kChY; d.lds = ldc;
            d.index = m_IDX; d.cond = m_COND; d.block_count = m_CNT;
            d.H = g.H16; d.W = g.W16; d.C = kChY; d.skip_thres = m_skip_thres;
            mask_dec_index(d, st);
            compact(m_IDX, 1, m_COND, m_CNT, ny, m_CIDX, m_TOTALS, 0, st);
        });
        // (one copy for count + indexes in front of the context network was measured SLOWER here - LD 300 -> 271,
        // HT-S 570 -> 541 pictures/s: the 192 KB copy sits in the stream in front of the context network's
        // kernels; the two small copies with a synchronisation in between do not hold them up)
        hip_check(hipMemcpyAsync(m_h_totals.get(), m_TOTALS, sizeof(int32_t), hipMemcpyDeviceToHost, st), "D2H totals");
        hip_check(hipStreamSynchronize(st), "sync");
        const int n = m_h_totals[0];
        if (n > 0) {
            hip_check(hipMemcpyAsync(m_h_idx.get(), m_CIDX, n, hipMemcpyDeviceToHost, st), "D2H indexes");
            hip_check(hipEventRecord(m_ev_idx, st), "hipEventRecord");
        }
        // context network + the y-independent reduction run while the host decodes y
        run_stage(kDec2, st, [&] {
            run_fe(st);
            run_reduction(st);
        });
        if (n > 0) {
            hip_check(hipEventSynchronize(m_ev_idx), "hipEventSynchronize");
            m_dec.decode_y(m_h_idx.get(), n, m_h_dec.get());
            hip_check(hipMemcpyAsync(m_DECODED, m_h_dec.get(), n, hipMemcpyHostToDevice, st), "H2D symbols");
        }
        run_stage(kDecStep + 3, st, [&] {
            MaskStepDec d;
            d.decoded = m_DECODED; d.cond = m_COND; d.block_count = m_CNT; d.totals = m_TOTALS; d.yq = m_YQ;
            d.q_dec = m_COMMON; d.ldq = ldc;
            d.y_hat = m_CATSP; d.ldh = 2 * kChY;
            d.H = g.H16; d.W = g.W16; d.C = kChY; d.nsteps = 4;
            for (int k = 0; k < 4; ++k) {
                if (k == 0) { d.means = m_COMMON + 2 * kChY; d.ldm = ldc; }
                else { run_spatial_prior(k - 1, st); d.means = m_SP; d.ldm = kChY; }
                d.step = k;
                mask_step_dec(d, st);
            }
            run_decoder(st);
            run_recon_head(x_hat, st);
        });
    } else {
        const int nq = g.P16() * (kChY / 4);
        auto index_step = [&](int k) {
            YStepDecIndex d;
            if (k == 0) { d.scales = m_COMMON + kChY; d.lds = ldc; }
            else { d.scales = m_SP; d.lds = 2 * kChY; }
            d.index = m_IDX; d.cond = m_COND; d.block_count = m_CNT;
            d.H = g.H16; d.W = g.W16; d.C = kChY; d.step = k; d.skip_thres = m_skip_thres;
            y_step_dec_index(d, st);
            uint8_t* region = m_CIDX + static_cast<size_t>(k) * m_idx_region;
            compact(m_IDX, 1, m_COND, m_CNT, nq, region + 16, reinterpret_cast<int32_t*>(region), 0, st);
        };
        run_stage(kDec1, st, [&] {
            int8_to_half(m_ZI8, m_ZH, nz, st);
            run_common(st);
            run_reduction(st);
            run_fe(st);
            index_step(0);
        });
        for (int k = 0; k < 4; ++k) {
            const uint8_t* region = m_CIDX + static_cast<size_t>(k) * m_idx_region;
            uint8_t* h_region = m_h_idx.get() + static_cast<size_t>(k) * m_idx_region;
            const size_t first = std::min(m_idx_region, 16 + kFirstIdxCopy);
            hip_check(hipMemcpyAsync(h_region, region, first, hipMemcpyDeviceToHost, st), "D2H count + indexes");
            hip_check(hipStreamSynchronize(st), "sync");
            const int n = *reinterpret_cast<const int32_t*>(h_region);
            if (n < 0 || n > nq) throw std::runtime_error("DMC-HT decompress: bad symbol count");
            if (16 + static_cast<size_t>(n) > first) {
                hip_check(hipMemcpyAsync(h_region + first, region + first, 16 + n - first, hipMemcpyDeviceToHost, st), "D2H indexes");
                hip_check(hipStreamSynchronize(st), "sync");
            }
            int8_t* h_dec = m_h_dec.get() + static_cast<size_t>(k) * nq;
            if (n > 0) {
                m_dec.decode_y(h_region + 16, n, h_dec);
                hip_check(hipMemcpyAsync(m_DECODED + static_cast<size_t>(k) * nq, h_dec, n, hipMemcpyHostToDevice, st), "H2D symbols");
            }
            run_stage(kDecStep + k, st, [&] {
                YStepDecRestore d;
                d.decoded = m_DECODED + static_cast<size_t>(k) * nq; d.cond = m_COND; d.block_count = m_CNT; d.totals = m_TOTALS; d.slot = 0;
                if (k == 0) { d.means = m_COMMON + 2 * kChY; d.ldm = ldc; }
                else { d.means = m_SP + kChY; d.ldm = 2 * kChY; }
                d.y_hat_acc = m_CATSP; d.ldacc = 2 * kChY;
                d.H = g.H16; d.W = g.W16; d.C = kChY; d.step = k; d.first = (k == 0);
                y_step_dec_restore(d, st);
                if (k < 3) {
                    run_spatial_prior(k, st);
                    index_step(k + 1);
                } else {
                    scale_clamped(m_CATSP, 2 * kChY, m_COMMON, ldc, m_CATSP, 2 * kChY, g.P16(), kChY, false, st);
                    run_decoder(st);
                    run_recon_head(x_hat, st);
                }
            });
        }
    }
    leave(user);
    m_has_feature_p = true;
    m_has_ref = true;
    m_memory_has_value = !reset;
    m_enc_ready = false;                   // ctx was advanced by this chunk; the encoder side must be re-seeded
}

// ------------------------------------------------------------------------------------ recon-head fan-out
void DmcHtCodec::set_recon_mask(unsigned mask)
{
    mask &= 0xffu;
    if (mask == m_recon_mask) return;
    quiesce();
    clear_graphs();              // the captured decode stages contain the head launches
    m_recon_mask = mask;
}

size_t DmcHtCodec::export_feature(void* dst, size_t cap, hipStream_t user)
{
    if (!m_has_params || m_g.H8 == 0 || !m_has_feature_p) throw std::runtime_error("DMC-HT export_feature: no feature_p yet");
    const Geometry& g = m_g;
    const size_t bytes = static_cast<size_t>(g.P8()) * kChD * sizeof(half_t);
    if (dst == nullptr) return bytes;
    if (cap < bytes) throw std::invalid_argument("DMC-HT export_feature: destination too small");
    hipStream_t st = enter(user);
    // strided [P8][1024] view -> dense [P8][512]
    replicate_pad(m_CATM + kChM, kChM + kChD, g.H8, g.W8, kChD, 0, 0, static_cast<half_t*>(dst), kChD, st);
    leave(user);
    return bytes;
}

void DmcHtCodec::import_feature(const void* src, size_t bytes, int height, int width, hipStream_t user)
{
    prepare(height, width);
    const Geometry& g = m_g;
    if (bytes != static_cast<size_t>(g.P8()) * kChD * sizeof(half_t)) {
        throw std::invalid_argument("DMC-HT import_feature: size does not match the picture size");
    }
    hipStream_t st = enter(user);
    replicate_pad(static_cast<const half_t*>(src), kChD, g.H8, g.W8, kChD, 0, 0, m_CATM + kChM, kChM + kChD, st);
    leave(user);
    m_has_feature_p = true;
}

void DmcHtCodec::run_recon_heads(unsigned mask, half_t* x_hat, hipStream_t user)
{
    if (!m_has_params || m_g.H8 == 0 || !m_has_feature_p) throw std::runtime_error("DMC-HT run_recon_heads: no feature_p");
    hipStream_t st = enter(user);
    const unsigned keep = m_recon_mask;
    m_recon_mask = mask & 0xffu;
    try {
        run_recon_head(x_hat, st);           // eager: not part of a captured stage
    } catch (...) {
        m_recon_mask = keep;
        throw;
    }
    m_recon_mask = keep;
    leave(user);
}

// ------------------------------------------------------------------------------------ state hand-off
namespace {

struct HtStateHeader {
    uint32_t magic, h8, w8, flags;
    uint32_t reserved[12];
};
constexpr uint32_t kHtMagic = 0x44434854;     // "DCHT"

}  // namespace

size_t DmcHtCodec::export_state(void* dst, size_t cap, hipStream_t user)
{
    if (!m_has_params || m_g.H8 == 0) throw std::runtime_error("DMC-HT export_state: no state yet");
    const Geometry& g = m_g;
    const size_t n_fi = static_cast<size_t>(g.P8()) * kChSrcI, n_catm = static_cast<size_t>(g.P8()) * (kChM + kChD),
                 n_ctx = static_cast<size_t>(g.P8()) * kChD;
    const size_t bytes = sizeof(HtStateHeader) + 2 * (n_fi + n_catm + n_ctx);
    if (dst == nullptr) return bytes;
    if (cap < bytes) throw std::invalid_argument("export_state: destination too small");
    hipStream_t st = enter(user);
    HtStateHeader h{};
    h.magic = kHtMagic ^ (m_hts ? 1u : 0u); h.h8 = g.H8; h.w8 = g.W8;
    h.flags = (m_has_ref ? 1u : 0u) | (m_enc_ready ? 2u : 0u) | (m_memory_has_value ? 4u : 0u) | (m_has_feature_p ? 8u : 0u);
    char* out = static_cast<char*>(dst);
    hip_check(hipMemcpyAsync(out, &h, sizeof(h), hipMemcpyHostToDevice, st), "state header");
    hip_check(hipStreamSynchronize(st), "sync");
    out += sizeof(h);
    hip_check(hipMemcpyAsync(out, m_FI, 2 * n_fi, hipMemcpyDeviceToDevice, st), "state D2D");
    out += 2 * n_fi;
    hip_check(hipMemcpyAsync(out, m_CATM, 2 * n_catm, hipMemcpyDeviceToDevice, st), "state D2D");
    out += 2 * n_catm;
    hip_check(hipMemcpy2DAsync(out, 2 * kChD, m_CATE + kChSrc, 2 * (kChSrc + kChD), 2 * kChD, g.P8(), hipMemcpyDeviceToDevice, st), "state ctx");
    leave(user);
    return bytes;
}

void DmcHtCodec::import_state(const void* src, size_t bytes, int height, int width, hipStream_t user)
{
    prepare(height, width);
    const Geometry& g = m_g;
    const size_t n_fi = static_cast<size_t>(g.P8()) * kChSrcI, n_catm = static_cast<size_t>(g.P8()) * (kChM + kChD),
                 n_ctx = static_cast<size_t>(g.P8()) * kChD;
    if (bytes != sizeof(HtStateHeader) + 2 * (n_fi + n_catm + n_ctx)) {
        throw std::invalid_argument("import_state: size does not match this picture size");
    }
    hipStream_t st = enter(user);
    HtStateHeader h{};
    const char* in = static_cast<const char*>(src);
    hip_check(hipMemcpyAsync(&h, in, sizeof(h), hipMemcpyDeviceToHost, st), "state header");
    hip_check(hipStreamSynchronize(st), "sync");
    if (h.magic != (kHtMagic ^ (m_hts ? 1u : 0u)) || h.h8 != static_cast<uint32_t>(g.H8) || h.w8 != static_cast<uint32_t>(g.W8)) {
        throw std::invalid_argument("import_state: not a state of this model structure and picture size");
    }
    in += sizeof(h);
    hip_check(hipMemcpyAsync(m_FI, in, 2 * n_fi, hipMemcpyDeviceToDevice, st), "state D2D");
    in += 2 * n_fi;
    hip_check(hipMemcpyAsync(m_CATM, in, 2 * n_catm, hipMemcpyDeviceToDevice, st), "state D2D");
    in += 2 * n_catm;
    hip_check(hipMemcpy2DAsync(m_CATE + kChSrc, 2 * (kChSrc + kChD), in, 2 * kChD, 2 * kChD, g.P8(), hipMemcpyDeviceToDevice, st), "state ctx");
    leave(user);
    m_has_ref = h.flags & 1u; m_enc_ready = h.flags & 2u; m_memory_has_value = h.flags & 4u; m_has_feature_p = h.flags & 8u;
}

// ------------------------------------------------------------------------------------ debug
size_t DmcHtCodec::debug_read(const std::string& name, void* dst, size_t cap, hipStream_t st)
{
    const Geometry& g = m_g;
    const void* src = nullptr;
    size_t pixels = 0, ch_bytes = 0, pitch = 0;
    auto view = [&](const void* p, int P, int c, int ld, int elem) {
        src = p; pixels = P; ch_bytes = static_cast<size_t>(c) * elem; pitch = static_cast<size_t>(ld) * elem;
    };
    if (name == "y") view(m_Y, g.P16(), kChY, kChY, 2);
    else if (name == "y_hat") view(m_CATSP, g.P16(), kChY, 2 * kChY, 2);
    else if (name == "common") view(m_COMMON, g.P16(), 3 * kChY, 3 * kChY, 2);
    else if (name == "z_i8") view(m_ZI8, g.P64(), kChZ, kChZ, 1);
    else if (name == "memory") view(m_CATM, g.P8(), kChM, kChM + kChD, 2);
    else if (name == "feature_p") view(m_CATM + kChM, g.P8(), kChD, kChM + kChD, 2);
    else if (name == "ctx") view(m_CATE + kChSrc, g.P8(), kChD, kChSrc + kChD, 2);
    else if (name == "feature_i") view(m_FI, g.P8(), kChSrcI, kChSrcI, 2);
    else if (name == "symbols") view(m_COMP, 1, g.P16() * kChY, g.P16() * kChY, 2);
    else if (name == "totals") view(m_TOTALS, 1, 4, 4, 4);
    else throw std::invalid_argument("unknown debug tensor '" + name + "'");
    const size_t bytes = pixels * ch_bytes;
    if (dst != nullptr) {
        if (cap < bytes) throw std::invalid_argument("debug_read: destination too small");
        hip_check(hipStreamSynchronize(st), "sync");
        hip_check(hipStreamSynchronize(m_cs), "sync");
        hip_check(hipMemcpy2D(dst, ch_bytes, src, pitch, ch_bytes, pixels, hipMemcpyDeviceToHost), "debug D2H");
    }
    return bytes;
}

}  // namespace dcvc
