// DMC-LD codec orchestration on MI355X (see dmc_ld.h). Dataflow follows dmc_ld_proxy.cpp:407-593;
// as in the intra codec the launch structure is ours: one hipGraph per stage for all qps, the
// concatenations of the reference (torch.cat inputs of the adaptor blocks) are fixed channel
// ranges of resident buffers, and the two checkerboard steps are fused symbol kernels.
#include "codec/dmc_ld.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace dcvc {

namespace {

enum StageKey : int { kRef = 0, kEnc0 = 1, kEnc1 = 2 /* +reset */, kDec0 = 10 /* +memory_has_value */,
                      kDec1 = 12, kDec2 = 13, kDec3 = 14 };

}  // namespace

DmcLdCodec::DmcLdCodec()
{
    // LD's kernels are short (4.2 ms of kernel time in ~350 launches per 1080p picture): measured on
    // MI355X, eager launches on the codec's stream beat hipGraph replay (139 vs 123 pictures/s), the
    // other codecs are neutral. set_use_graphs(true) switches back.
    m_use_graphs = false;
    hip_check(hipEventCreateWithFlags(&m_ev_idx, hipEventDisableTiming), "hipEventCreate");
}

DmcLdCodec::~DmcLdCodec()
{
    quiesce();
    if (m_ev_idx) (void)hipEventDestroy(m_ev_idx);
}

// ------------------------------------------------------------------------------------ set_param
void DmcLdCodec::set_param(const ParamStore& ps, float skip_thres)
{
    quiesce();
    clear_graphs();
    m_wmem.release();
    kernels_init();
    m_skip_thres = skip_thres;
    m_q_encoder = upload_qp_table(ps, m_wmem, "q_encoder", kChD);
    m_q_decoder = upload_qp_table(ps, m_wmem, "q_decoder", kChD);
    m_q_feature = upload_qp_table(ps, m_wmem, "q_feature", 2 * kChY);
    m_cur_q_encoder = m_wmem.alloc_half(kChD);
    m_cur_q_decoder = m_wmem.alloc_half(kChD);
    m_cur_q_feature = m_wmem.alloc_half(2 * kChY);
    m_zeros = m_wmem.alloc_half(2048);
    auto chain = [&](DcbW* blocks, int n, const std::string& prefix) {
        for (int i = 0; i < n; ++i) blocks[i].load(ps, m_wmem, prefix + std::to_string(i) + ".");
    };
    chain(m_fa_i, 4, "feature_adaptor_i.conv.");
    chain(m_fa_m, 4, "feature_adaptor_m.conv.");
    chain(m_fe, 5, "feature_extractor.conv.");
    chain(m_encb, 2, "encoder.conv1.");
    m_encb[2].load(ps, m_wmem, "encoder.conv2.");
    m_enc_down.load(ps, m_wmem, "encoder.down.");
    m_henc0.load(ps, m_wmem, "hyper_encoder.conv.0.");
    m_henc1.load(ps, m_wmem, "hyper_encoder.conv.1.", false);      // dmc_ld_proxy.cpp:250-251
    m_henc2.load(ps, m_wmem, "hyper_encoder.conv.2.", false);
    m_hdec0.load(ps, m_wmem, "hyper_decoder.conv.0.", false);      // dmc_ld_proxy.cpp:221-222
    m_hdec1.load(ps, m_wmem, "hyper_decoder.conv.1.", false);
    m_hdec2.load(ps, m_wmem, "hyper_decoder.conv.2.");
    m_tpe.load(ps, m_wmem, "temporal_prior_encoder.conv.", false); // dmc_ld_proxy.cpp:366
    chain(m_fus, 3, "y_prior_fusion.conv.");
    m_fus3.load(ps, m_wmem, "y_prior_fusion.conv.3.");
    chain(m_sp, 2, "y_spatial_prior.conv.");
    m_sp2.load(ps, m_wmem, "y_spatial_prior.conv.2.");
    m_dec_up.load(ps, m_wmem, "decoder.up.");
    chain(m_dec1, 3, "decoder.conv1.");
    m_dec2.load(ps, m_wmem, "decoder.conv2.");
    chain(m_rh, 3, "recon_head.conv.");
    m_rh_head.load(ps, m_wmem, "recon_head.head.");
    load_cdf_tables(ps);
    m_has_params = true;
    m_has_ref = m_enc_ready = m_memory_has_value = m_has_feature_p = false;
}

// ------------------------------------------------------------------------------------ buffers
void DmcLdCodec::prepare(int height, int width)
{
    if (!m_has_params) throw std::runtime_error("DMC-LD: set_param() has not been called");
    if (height <= 0 || width <= 0) throw std::invalid_argument("DMC-LD: empty picture");
    const int H8 = ceil_div(height, 16) * 2, W8 = ceil_div(width, 16) * 2;
    if (m_g.H8 == H8 && m_g.W8 == W8) return;
    quiesce();
    clear_graphs();
    m_bmem.release();
    m_has_ref = m_enc_ready = m_memory_has_value = m_has_feature_p = false;   // the state had another size
    Geometry g;
    g.H8 = H8; g.W8 = W8;
    g.H16 = H8 / 2; g.W16 = W8 / 2;
    g.H16p = ceil_div(g.H16, 4) * 4; g.W16p = ceil_div(g.W16, 4) * 4;      // dmc_common.cpp:73-83
    g.H32 = g.H16p / 2; g.W32 = g.W16p / 2;
    g.H64 = g.H16p / 4; g.W64 = g.W16p / 4;
    m_g = g;
    auto H = [&](size_t n) { return m_bmem.alloc_half(n); };
    const size_t P8 = g.P8(), P16 = g.P16(), P16p = g.P16p(), P32 = g.P32(), P64 = g.P64();
    m_s.elems = std::max<size_t>(P8 * (kChM / 2), P16p * (3 * kChY / 2));
    m_s.t1 = H(m_s.elems); m_s.t2 = H(m_s.elems); m_s.t3 = H(m_s.elems);
    m_FI = H(P8 * kChSrc);
    m_CATM = H(P8 * (kChM + kChD));
    m_CATD = H(P8 * (kChD + kChM));
    m_T = H(P8 * kChM); m_T2 = H(P8 * kChM);
    m_Y = H(P16 * kChY); m_Ypad = g.padded() ? H(P16p * kChY) : m_Y;
    m_Z1 = H(P16p * kChZ); m_Z2 = H(P32 * kChZ); m_Z3 = H(P64 * kChZ); m_ZH = H(P64 * kChZ);
    m_ZI8 = static_cast<int8_t*>(m_bmem.alloc(P64 * kChZ));
    m_H1 = H(P32 * kChZ); m_H2 = H(P16p * kChZ); m_HP = H(P16p * kChY);
    m_CATPF = H(P16 * 3 * kChY);
    m_CATSP = H(P16 * 4 * kChY);
    m_SPT = H(P16 * 2 * kChY);
    m_MEANS1 = H(P16 * kChY);
    const size_t n = P16 * kChY;
    m_SYM = static_cast<int16_t*>(m_bmem.alloc(n * 2));
    m_COMP = static_cast<int16_t*>(m_bmem.alloc(n * 2));
    m_COND = static_cast<uint8_t*>(m_bmem.alloc(n / 8 + 8));
    m_IDX = static_cast<uint8_t*>(m_bmem.alloc(n));
    m_CIDX = static_cast<uint8_t*>(m_bmem.alloc(n));
    m_DECODED = static_cast<int8_t*>(m_bmem.alloc(n));
    m_YQ = static_cast<int8_t*>(m_bmem.alloc(n));
    m_CNT = static_cast<int32_t*>(m_bmem.alloc(sizeof(int32_t) * symbol_blocks(static_cast<int>(n))));
    m_TOTALS = static_cast<int32_t*>(m_bmem.alloc(sizeof(int32_t) * 4));
    m_h_totals.reserve(16);
    m_h_sym.reserve(n);
    m_h_z.reserve(P64 * kChZ + 64);
    m_h_idx.reserve(n);
    m_h_dec.reserve(n);
}

void DmcLdCodec::select_qp(int qp, hipStream_t st)
{
    copy_qp_rows({{m_cur_q_encoder, m_q_encoder, kChD}, {m_cur_q_decoder, m_q_decoder, kChD},
                  {m_cur_q_feature, m_q_feature, 2 * kChY}}, qp, st);
}

// ------------------------------------------------------------------------------------ networks
void DmcLdCodec::run_fa_i(hipStream_t st, bool feed_fe)
{
    const View t(m_T, kChM, kChM);
    const bool feed = feed_fe && m_fa_i[3].feeds(m_fe[0]);
    run_dcb_chain(m_fa_i, 4, View(m_FI, kChSrc, kChSrc), t, View(m_CATM, kChM + kChD, kChM), m_g.H8, m_g.W8, m_s, st,
                  nullptr, View(m_T2, kChM, kChM), nullptr, feed ? &m_fe[0] : nullptr);
}

void DmcLdCodec::run_fa_m(hipStream_t st, bool feed_fe)
{
    const View t(m_T, kChM, kChM);
    const bool feed = feed_fe && m_fa_m[3].feeds(m_fe[0]);
    run_dcb_chain(m_fa_m, 4, View(m_CATM, kChM + kChD, kChM + kChD), t, View(m_CATM, kChM + kChD, kChM),
                  m_g.H8, m_g.W8, m_s, st, nullptr, View(m_T2, kChM, kChM), nullptr, feed ? &m_fe[0] : nullptr);
}

void DmcLdCodec::run_fe(hipStream_t st, bool dc0_done)
{
    const View t(m_T, kChM, kChM);
    run_dcb_chain(m_fe, 5, View(m_CATM, kChM + kChD, kChM), t, View(m_CATD + kChD, kChD + kChM, kChM),
                  m_g.H8, m_g.W8, m_s, st, nullptr, View(m_T2, kChM, kChM), nullptr, nullptr, dc0_done);
}

void DmcLdCodec::run_tpe(hipStream_t st)
{
    const View out(m_CATPF + kChY, 3 * kChY, 2 * kChY);
    m_tpe.forward(View(m_CATM, kChM + kChD, kChM), out, out, m_g.H8, m_g.W8, m_zeros, m_s, st);
}

void DmcLdCodec::run_encoder(hipStream_t st)
{
    const Geometry& g = m_g;
    const View t(m_T, kChD, kChD);
    // [x unshuffled | ctx] = channels 64..511 of CATD (dmc_ld_proxy.cpp:755-757)
    // encoder.conv1 (two blocks) and encoder.conv2 (its output scaled by q_encoder inside its last conv) as ONE chain: every
    // launch also computes dc.0 of the block behind it
    run_dcb_chain(m_encb, 3, View(m_CATD + 64, kChD + kChM, kChSrc + kChM), t, t, g.H8, g.W8, m_s, st, m_cur_q_encoder,
                  View(m_T2, kChD, kChD));
    ConvKxKDesc d;
    d.x = m_T; d.ldx = kChD; d.w = m_enc_down.w; d.bias = m_enc_down.b; d.zeros = m_zeros;
    d.y = m_Y; d.ldy = kChY; d.in_h = g.H8; d.in_w = g.W8; d.cin = kChD; d.cout = kChY;
    d.ksize = 3; d.stride = 2; d.pad = 1;
    conv_kxk(d, st);
}

void DmcLdCodec::run_hyper_encoder(hipStream_t st)
{
    const Geometry& g = m_g;
    if (g.padded()) {
        replicate_pad(m_Y, kChY, g.H16, g.W16, kChY, g.H16p - g.H16, g.W16p - g.W16, m_Ypad, kChY, st);
    }
    const View z1(m_Z1, kChZ, kChZ), z2(m_Z2, kChZ, kChZ), z3(m_Z3, kChZ, kChZ);
    m_henc0.forward(View(m_Ypad, kChY, kChY), z1, g.H16p, g.W16p, m_s, st);
    m_henc1.forward(z1, z2, z2, g.H16p, g.W16p, m_zeros, m_s, st);
    m_henc2.forward(z2, z3, z3, g.H32, g.W32, m_zeros, m_s, st);
    round_z(m_Z3, m_ZH, m_ZI8, g.P64() * kChZ, st);
}

void DmcLdCodec::run_priors(hipStream_t st)
{
    const Geometry& g = m_g;
    const View h1(m_H1, kChZ, kChZ), h2(m_H2, kChZ, kChZ);
    m_hdec0.forward(View(m_ZH, kChZ, kChZ), h1, h1, g.H64, g.W64, m_s, st);
    m_hdec1.forward(h1, h2, h2, g.H32, g.W32, m_s, st);
    m_hdec2.forward(h2, View(m_HP, kChY, kChY), g.H16p, g.W16p, m_s, st);
    // crop_hyper_params -> first third of the fusion input (dmc_ld_proxy.cpp:437-438)
    crop(m_HP, kChY, g.W16p, m_CATPF, 3 * kChY, g.H16, g.W16, kChY, st);
    mul_channel(m_CATPF + kChY, 3 * kChY, m_cur_q_feature, m_CATPF + kChY, 3 * kChY, g.P16(), 2 * kChY, st);
    const View pf(m_CATPF, 3 * kChY, 3 * kChY);
    // y_prior_fusion: three (384, 192) blocks in place, each launch with dc.0 of the next inside, the last one with
    // y_prior_fusion.conv.3 -> (q_dec | scales | means) behind y_hat in the spatial-prior input
    const FinCall fin(m_fus3, m_CATSP + kChY, 4 * kChY);
    run_dcb_chain(m_fus, 3, pf, pf, pf, g.H16, g.W16, m_s, st, nullptr, View(), &fin);
}

void DmcLdCodec::run_spatial_prior(hipStream_t st)
{
    const Geometry& g = m_g;
    const View t(m_SPT, 2 * kChY, 2 * kChY);
    const FinCall fin(m_sp2, m_MEANS1, kChY);              // y_spatial_prior.conv.2 closes the chain
    run_dcb_chain(m_sp, 2, View(m_CATSP, 4 * kChY, 4 * kChY), t, t, g.H16, g.W16, m_s, st, nullptr, View(), &fin);
}

void DmcLdCodec::run_decoder(hipStream_t st)
{
    const Geometry& g = m_g;
    m_dec_up.forward(View(m_CATSP, 4 * kChY, kChY), View(m_CATD, kChD + kChM, kChD), g.H16, g.W16, st);
    const View t(m_T, kChD, kChD);
    // decoder.conv2 (conv1x1_bias_with_quant) closes the chain -> feature_p, second half of the adaptor_m input
    const FinCall fin(m_dec2, m_CATM + kChM, kChM + kChD, m_cur_q_decoder);
    run_dcb_chain(m_dec1, 3, View(m_CATD, kChD + kChM, kChD + kChM), t, t, g.H8, g.W8, m_s, st, nullptr,
                  View(m_T2, kChD, kChD), &fin);
}

void DmcLdCodec::run_recon_head(half_t* x_hat, hipStream_t st)
{
    const Geometry& g = m_g;
    const View t(m_T, kChD, kChD);
    const FinCall fin(m_rh_head, m_FI, kChSrc);            // the head output doubles as the reference feature after a reset
    run_dcb_chain(m_rh, 3, View(m_CATM + kChM, kChM + kChD, kChD), t, t, g.H8, g.W8, m_s, st, nullptr,
                  View(m_T2, kChD, kChD), &fin);
    if (x_hat != nullptr) shuffle8(m_FI, kChSrc, g.H8, g.W8, 3, true, x_hat, st);
}

// ------------------------------------------------------------------------------------ reference frame
void DmcLdCodec::add_ref_feature_from_frame(const half_t* frame, int height, int width,
                                            bool apply_adaptor, hipStream_t user)
{
    prepare(height, width);
    hipStream_t st = enter(user);
    pad_unshuffle8(frame, height, width, 3, m_FI, m_g.H8, m_g.W8, st);
    if (apply_adaptor) {
        run_stage(kRef, st, [&] {
            run_fa_i(st, true);
            run_fe(st, m_fa_i[3].feeds(m_fe[0]));
            run_tpe(st);
        });
    }
    leave(user);
    m_has_ref = true;
    m_enc_ready = apply_adaptor;
    m_memory_has_value = apply_adaptor;
    m_has_feature_p = false;
}

// ------------------------------------------------------------------------------------ compress
int DmcLdCodec::compress(const half_t* x, int height, int width, int qp, bool reset, hipStream_t user)
{
    prepare(height, width);
    if (!m_enc_ready) {
        throw std::runtime_error("DMC-LD compress: no reference feature "
                                 "(call add_ref_feature_from_frame(frame, true) first)");
    }
    const Geometry& g = m_g;
    hipStream_t st = enter(user);
    select_qp(qp, st);
    pad_unshuffle8(x, height, width, 3, m_CATD + 64, g.H8, g.W8, st, kChD + kChM);   // x varies: outside the graph
    run_stage(kEnc0, st, [&] {
        run_encoder(st);
        run_hyper_encoder(st);
        run_priors(st);
        MaskStepEnc d;
        d.y = m_Y; d.ldy = kChY;
        d.q_dec = m_CATSP + kChY; d.ldq = 4 * kChY;
        d.scales = m_CATSP + 2 * kChY; d.lds = 4 * kChY;
        d.means = m_CATSP + 3 * kChY; d.ldm = 4 * kChY;
        d.y_hat = m_CATSP; d.ldh = 4 * kChY;
        d.sym = m_SYM; d.cond = m_COND; d.block_count = m_CNT;
        d.H = g.H16; d.W = g.W16; d.C = kChY; d.step = 0; d.skip_thres = m_skip_thres;
        mask_step_enc(d, st);
        run_spatial_prior(st);
        d.means = m_MEANS1; d.ldm = kChY; d.step = 1;
        mask_step_enc(d, st);
        compact(m_SYM, 2, m_COND, m_CNT, g.P16() * kChY, m_COMP, m_TOTALS, 0, st);
    });
    submit(st, [this, qp] { entropy_encode(qp); });
    // decoder + temporal state update run on the GPU while the worker entropy-codes on the host
    run_stage(kEnc1 + (reset ? 1 : 0), st, [&] {
        run_decoder(st);
        if (reset) {
            run_recon_head(nullptr, st);     // forward_reset, dmc_ld_proxy.cpp:297-305
            run_fa_i(st, true);
        } else {
            run_fa_m(st, true);
        }
        run_fe(st, (reset ? m_fa_i[3] : m_fa_m[3]).feeds(m_fe[0]));      // its first dc.0 came with the adaptor chain's last launch
        run_tpe(st);
    });
    leave(user);
    m_has_feature_p = true;
    wait_job();
    return m_ec_parallel;
}

void DmcLdCodec::entropy_encode(int qp)
{
    static const bool trace = getenv("DCVC_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (trace) {
            fprintf(stderr, "[dcvc]   ld encode: %s at +%.0f us\n", what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
    };
    // dmc_ld_proxy.cpp worker, Encode: y symbols then z
    const Geometry& g = m_g;
    hip_check(hipMemcpyAsync(m_h_totals.get(), m_TOTALS, sizeof(int32_t), hipMemcpyDeviceToHost, m_io_stream), "D2H totals");
    const int nz = g.P64() * kChZ;
    hip_check(hipMemcpyAsync(m_h_z.get(), m_ZI8, nz, hipMemcpyDeviceToHost, m_io_stream), "D2H z");
    hip_check(hipStreamSynchronize(m_io_stream), "sync io");
    const int total = m_h_totals[0];
    lap("totals + z on the host (GPU stage 0 done)");
    if (total > 0) {
        hip_check(hipMemcpyAsync(m_h_sym.get(), m_COMP, static_cast<size_t>(total) * 2, hipMemcpyDeviceToHost, m_io_stream), "D2H symbols");
        hip_check(hipStreamSynchronize(m_io_stream), "sync io");
    }
    lap("symbols on the host");
    m_ec_parallel = ec_parallel_for(total);
    m_enc.reset();
    m_enc.set_parallel(m_ec_parallel);
    m_enc.push_y(m_h_sym.get(), total);
    m_enc.push_z(m_h_z.get(), nz, qp * kChZ, kChZ);
    m_enc.flush();
    lap("rANS done");
}

// ------------------------------------------------------------------------------------ decompress
void DmcLdCodec::decompress(const uint8_t* bits, size_t nbytes, int qp, int height, int width,
                            int ec_parallel, bool reset, half_t* x_hat, hipStream_t user)
{
    prepare(height, width);
    if (m_memory_has_value ? !m_has_feature_p : !m_has_ref) {
        throw std::runtime_error("DMC-LD decompress: no reference feature "
                                 "(call add_ref_feature_from_frame first)");
    }
    const Geometry& g = m_g;
    hipStream_t st = enter(user);
    select_qp(qp, st);
    const bool extend = m_memory_has_value;
    run_stage(kDec0 + (extend ? 1 : 0), st, [&] {
        if (extend) run_fa_m(st);
        else run_fa_i(st);
        run_tpe(st);
    });
    // z is decoded on the host while the GPU updates the feature memory
    m_dec.set_parallel(ec_parallel);
    m_dec.set_stream(bits, nbytes);
    const int nz = g.P64() * kChZ;
    const int ny = g.P16() * kChY;
    m_dec.decode_z(nz, qp * kChZ, kChZ, m_h_z.get());
    hip_check(hipMemcpyAsync(m_ZI8, m_h_z.get(), nz, hipMemcpyHostToDevice, st), "H2D z");
    run_stage(kDec1, st, [&] {
        int8_to_half(m_ZI8, m_ZH, nz, st);
        run_priors(st);
        MaskDecIndex d;
        d.scales = m_CATSP + 2 * kChY; d.lds = 4 * kChY;
        d.index = m_IDX; d.cond = m_COND; d.block_count = m_CNT;
        d.H = g.H16; d.W = g.W16; d.C = kChY; d.skip_thres = m_skip_thres;
        mask_dec_index(d, st);
        compact(m_IDX, 1, m_COND, m_CNT, ny, m_CIDX, m_TOTALS, 0, st);
    });
    hip_check(hipMemcpyAsync(m_h_totals.get(), m_TOTALS, sizeof(int32_t), hipMemcpyDeviceToHost, st), "D2H totals");
    hip_check(hipStreamSynchronize(st), "sync");
    const int n = m_h_totals[0];
    if (n > 0) {
        hip_check(hipMemcpyAsync(m_h_idx.get(), m_CIDX, n, hipMemcpyDeviceToHost, st), "D2H indexes");
        hip_check(hipEventRecord(m_ev_idx, st), "hipEventRecord");
    }
    // the context network runs while the host decodes y (dmc_ld_proxy.cpp:556-560)
    run_stage(kDec2, st, [&] { run_fe(st); });
    if (n > 0) {
        hip_check(hipEventSynchronize(m_ev_idx), "hipEventSynchronize");
        m_dec.decode_y(m_h_idx.get(), n, m_h_dec.get());
        hip_check(hipMemcpyAsync(m_DECODED, m_h_dec.get(), n, hipMemcpyHostToDevice, st), "H2D symbols");
    }
    bind_stage_arg(kDec3, x_hat);
    run_stage(kDec3, st, [&] {
        MaskStepDec d;
        d.decoded = m_DECODED; d.cond = m_COND; d.block_count = m_CNT; d.totals = m_TOTALS; d.yq = m_YQ;
        d.means = m_CATSP + 3 * kChY; d.ldm = 4 * kChY;
        d.q_dec = m_CATSP + kChY; d.ldq = 4 * kChY;
        d.y_hat = m_CATSP; d.ldh = 4 * kChY;
        d.H = g.H16; d.W = g.W16; d.C = kChY; d.step = 0;
        mask_step_dec(d, st);
        run_spatial_prior(st);
        d.means = m_MEANS1; d.ldm = kChY; d.step = 1;
        mask_step_dec(d, st);
        run_decoder(st);
        run_recon_head(x_hat, st);
    });
    leave(user);
    m_has_feature_p = true;
    m_has_ref = true;                      // the recon-head output is the reference feature after a reset
    m_memory_has_value = !reset;
    m_enc_ready = false;                   // the temporal params were consumed by this picture
}

// ------------------------------------------------------------------------------------ state hand-off
namespace {

struct StateHeader {
    uint32_t magic, h8, w8, flags;
    uint32_t reserved[12];
};
constexpr uint32_t kLdMagic = 0x44434c44;     // "DCLD"

}  // namespace

size_t DmcLdCodec::export_state(void* dst, size_t cap, hipStream_t user)
{
    if (!m_has_params || m_g.H8 == 0) throw std::runtime_error("DMC-LD export_state: no state yet");
    const Geometry& g = m_g;
    const size_t n_fi = static_cast<size_t>(g.P8()) * kChSrc, n_catm = static_cast<size_t>(g.P8()) * (kChM + kChD),
                 n_ctx = static_cast<size_t>(g.P8()) * kChM, n_tp = static_cast<size_t>(g.P16()) * 2 * kChY;
    const size_t bytes = sizeof(StateHeader) + 2 * (n_fi + n_catm + n_ctx + n_tp);
    if (dst == nullptr) return bytes;
    if (cap < bytes) throw std::invalid_argument("export_state: destination too small");
    hipStream_t st = enter(user);
    StateHeader h{};
    h.magic = kLdMagic; h.h8 = g.H8; h.w8 = g.W8;
    h.flags = (m_has_ref ? 1u : 0u) | (m_enc_ready ? 2u : 0u) | (m_memory_has_value ? 4u : 0u) | (m_has_feature_p ? 8u : 0u);
    char* out = static_cast<char*>(dst);
    hip_check(hipMemcpyAsync(out, &h, sizeof(h), hipMemcpyHostToDevice, st), "state header");
    hip_check(hipStreamSynchronize(st), "sync");          // the header lives on this stack frame
    out += sizeof(h);
    auto put = [&](const half_t* p, size_t n) {
        hip_check(hipMemcpyAsync(out, p, 2 * n, hipMemcpyDeviceToDevice, st), "state D2D");
        out += 2 * n;
    };
    put(m_FI, n_fi);
    put(m_CATM, n_catm);
    // ctx and the temporal prior are channel slices of wider buffers: dense copies
    hip_check(hipMemcpy2DAsync(out, 2 * kChM, m_CATD + kChD, 2 * (kChD + kChM), 2 * kChM, g.P8(), hipMemcpyDeviceToDevice, st), "state ctx");
    out += 2 * n_ctx;
    hip_check(hipMemcpy2DAsync(out, 4 * kChY, m_CATPF + kChY, 6 * kChY, 4 * kChY, g.P16(), hipMemcpyDeviceToDevice, st), "state temporal");
    leave(user);
    return bytes;
}

void DmcLdCodec::import_state(const void* src, size_t bytes, int height, int width, hipStream_t user)
{
    prepare(height, width);
    const Geometry& g = m_g;
    const size_t n_fi = static_cast<size_t>(g.P8()) * kChSrc, n_catm = static_cast<size_t>(g.P8()) * (kChM + kChD),
                 n_ctx = static_cast<size_t>(g.P8()) * kChM, n_tp = static_cast<size_t>(g.P16()) * 2 * kChY;
    if (bytes != sizeof(StateHeader) + 2 * (n_fi + n_catm + n_ctx + n_tp)) {
        throw std::invalid_argument("import_state: size does not match this picture size");
    }
    hipStream_t st = enter(user);
    StateHeader h{};
    const char* in = static_cast<const char*>(src);
    hip_check(hipMemcpyAsync(&h, in, sizeof(h), hipMemcpyDeviceToHost, st), "state header");
    hip_check(hipStreamSynchronize(st), "sync");
    if (h.magic != kLdMagic || h.h8 != static_cast<uint32_t>(g.H8) || h.w8 != static_cast<uint32_t>(g.W8)) {
        throw std::invalid_argument("import_state: not a DMC-LD state of this picture size");
    }
    in += sizeof(h);
    auto get = [&](half_t* p, size_t n) {
        hip_check(hipMemcpyAsync(p, in, 2 * n, hipMemcpyDeviceToDevice, st), "state D2D");
        in += 2 * n;
    };
    get(m_FI, n_fi);
    get(m_CATM, n_catm);
    hip_check(hipMemcpy2DAsync(m_CATD + kChD, 2 * (kChD + kChM), in, 2 * kChM, 2 * kChM, g.P8(), hipMemcpyDeviceToDevice, st), "state ctx");
    in += 2 * n_ctx;
    hip_check(hipMemcpy2DAsync(m_CATPF + kChY, 6 * kChY, in, 4 * kChY, 4 * kChY, g.P16(), hipMemcpyDeviceToDevice, st), "state temporal");
    leave(user);
    m_has_ref = h.flags & 1u; m_enc_ready = h.flags & 2u; m_memory_has_value = h.flags & 4u; m_has_feature_p = h.flags & 8u;
}

// ------------------------------------------------------------------------------------ debug
size_t DmcLdCodec::debug_read(const std::string& name, void* dst, size_t cap, hipStream_t st)
{
    const Geometry& g = m_g;
    const void* src = nullptr;
    size_t pixels = 0, ch_bytes = 0, pitch = 0;
    auto view = [&](const void* p, int P, int c, int ld, int elem) {
        src = p; pixels = P; ch_bytes = static_cast<size_t>(c) * elem; pitch = static_cast<size_t>(ld) * elem;
    };
    if (name == "y") view(m_Y, g.P16(), kChY, kChY, 2);
    else if (name == "y_hat") view(m_CATSP, g.P16(), kChY, 4 * kChY, 2);
    else if (name == "common") view(m_CATSP + kChY, g.P16(), 3 * kChY, 4 * kChY, 2);
    else if (name == "means1") view(m_MEANS1, g.P16(), kChY, kChY, 2);
    else if (name == "z_i8") view(m_ZI8, g.P64(), kChZ, kChZ, 1);
    else if (name == "memory") view(m_CATM, g.P8(), kChM, kChM + kChD, 2);
    else if (name == "feature_p") view(m_CATM + kChM, g.P8(), kChD, kChM + kChD, 2);
    else if (name == "ctx") view(m_CATD + kChD, g.P8(), kChM, kChD + kChM, 2);
    else if (name == "temporal") view(m_CATPF + kChY, g.P16(), 2 * kChY, 3 * kChY, 2);
    else if (name == "feature_i") view(m_FI, g.P8(), kChSrc, kChSrc, 2);
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
