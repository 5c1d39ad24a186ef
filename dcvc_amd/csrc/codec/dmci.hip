// DMCI codec orchestration on MI355X (see dmci.h). Dataflow follows dmci_proxy.cpp:296-602; the
// launch structure does not: one hipGraph per stage for ALL qps (the per-qp scale vectors are
// copied into fixed device slots before the graph), fused per-step symbol kernels, a single
// device->host transfer of the compacted symbols of all four steps.
#include "codec/dmci.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>

namespace dcvc {

namespace {

enum StageKey : int { kEnc0 = 0, kEnc1 = 1, kDec0 = 10, kDec1 = 11 };

}  // namespace

DmciCodec::DmciCodec() = default;

DmciCodec::~DmciCodec()
{
    quiesce();
}

// ------------------------------------------------------------------------------------ set_param
void DmciCodec::set_param(const ParamStore& ps, float skip_thres)
{
    quiesce();            // compress() returns before its reconstruction graph has finished
    clear_graphs();
    m_wmem.release();
    kernels_init();
    m_skip_thres = skip_thres;
    m_q_enc = upload_qp_table(ps, m_wmem, "q_scale_enc", kChEncDec);
    m_q_dec = upload_qp_table(ps, m_wmem, "q_scale_dec", kChEncDec);
    m_q_y_enc = upload_qp_table(ps, m_wmem, "q_scale_y_enc", kChY);
    m_q_y_dec = upload_qp_table(ps, m_wmem, "q_scale_y_dec", kChY);
    m_zeros = m_wmem.alloc_half(2048);
    m_cur_q_enc = m_wmem.alloc_half(kChEncDec);
    m_cur_q_dec = m_wmem.alloc_half(kChEncDec);
    m_cur_q_y_enc = m_wmem.alloc_half(kChY);
    m_cur_q_y_dec = m_wmem.alloc_half(kChY);

    m_enc1.load(ps, m_wmem, "enc.enc_1.");
    for (int i = 0; i < 6; ++i) m_enc2[i].load(ps, m_wmem, "enc.enc_2." + std::to_string(i) + ".");
    m_enc_down.load(ps, m_wmem, "enc.enc_2.6.");
    m_henc0.load(ps, m_wmem, "hyper_enc.conv.0.");
    m_henc1.load(ps, m_wmem, "hyper_enc.conv.1.");
    m_henc2.load(ps, m_wmem, "hyper_enc.conv.2.");
    m_hdec0.load(ps, m_wmem, "hyper_dec.conv.0.");
    m_hdec1.load(ps, m_wmem, "hyper_dec.conv.1.");
    m_hdec2.load(ps, m_wmem, "hyper_dec.conv.2.");
    for (int i = 0; i < 3; ++i) m_fus[i].load(ps, m_wmem, "y_prior_fusion.conv." + std::to_string(i) + ".");
    m_fus3.load(ps, m_wmem, "y_prior_fusion.conv.3.");
    m_reduction.load(ps, m_wmem, "y_spatial_prior_reduction.");
    for (int i = 0; i < 3; ++i) {
        m_sp_adaptor[i].load(ps, m_wmem, "y_spatial_prior_adaptor_" + std::to_string(i + 1) + ".");
        m_sp[i].load(ps, m_wmem, "y_spatial_prior.conv." + std::to_string(i) + ".");
    }
    m_sp3.load(ps, m_wmem, "y_spatial_prior.conv.3.");
    m_dec_up.load(ps, m_wmem, "dec.dec_1.0.");
    for (int i = 0; i < 12; ++i) m_dec1[i].load(ps, m_wmem, "dec.dec_1." + std::to_string(i + 1) + ".");
    m_dec2.load(ps, m_wmem, "dec.dec_2.");

    load_cdf_tables(ps);
    m_has_params = true;
}

// ------------------------------------------------------------------------------------ buffers
void DmciCodec::prepare(int height, int width)
{
    if (!m_has_params) throw std::runtime_error("DMCI: set_param() has not been called");
    if (m_g.H == height && m_g.W == width) return;
    quiesce();
    clear_graphs();
    m_bmem.release();
    Geometry g;
    g.H = height; g.W = width;
    g.H8 = ceil_div(height, 16) * 2; g.W8 = ceil_div(width, 16) * 2;
    g.H16 = g.H8 / 2; g.W16 = g.W8 / 2;
    g.H16p = ceil_div(g.H16, 4) * 4; g.W16p = ceil_div(g.W16, 4) * 4;      // dmc_common.cpp:73-83
    g.H32 = g.H16p / 2; g.W32 = g.W16p / 2;
    g.H64 = g.H16p / 4; g.W64 = g.W16p / 4;
    m_g = g;
    auto H = [&](size_t n) { return m_bmem.alloc_half(n); };
    const size_t P8 = g.P8(), P16 = g.P16(), P16p = g.P16p(), P32 = g.P32(), P64 = g.P64();
    m_s.elems = std::max<size_t>(P8 * kChEncDec, P16p * 2 * kChY);
    m_s.t1 = H(m_s.elems); m_s.t2 = H(m_s.elems); m_s.t3 = H(m_s.elems);
    m_U = H(P8 * kChSrc); m_F = H(P8 * kChEncDec);
    m_Y = H(P16 * kChY); m_Ypad = g.padded() ? H(P16p * kChY) : m_Y;
    m_Z1 = H(P16p * kChZ); m_Z2a = H(P32 * kChZ); m_Z2 = H(P32 * kChZ);
    m_Z3a = H(P64 * kChZ); m_Z3 = H(P64 * kChZ); m_ZH = H(P64 * kChZ);
    m_ZI8 = static_cast<int8_t*>(m_bmem.alloc(P64 * kChZ));
    m_H1a = H(P32 * kChZ); m_H1 = H(P32 * kChZ); m_H2a = H(P16p * kChZ); m_H2 = H(P16p * kChZ);
    m_HP = H(P16p * kChY);
    m_PF = H(P16p * 2 * kChY); m_PARAMSp = H(P16p * 2 * kChY);
    m_PARAMS = g.padded() ? H(P16 * 2 * kChY) : m_PARAMSp;
    m_CAT = H(P16 * 2 * kChY); m_AD = H(P16 * 2 * kChY); m_SP = H(P16 * 2 * kChY);
    m_YHAT = H(P16 * kChY);
    m_D0 = H(P8 * kChEncDec); m_D1 = H(P8 * kChEncDec); m_R = H(P8 * kChSrc);
    const size_t nq = P16 * (kChY / 4);               // symbols per autoregressive step
    m_SYM = static_cast<int16_t*>(m_bmem.alloc(nq * 2));
    m_COMP = static_cast<int16_t*>(m_bmem.alloc(4 * nq * 2));
    m_COND = static_cast<uint8_t*>(m_bmem.alloc(nq / 8 + 8));
    m_IDX = static_cast<uint8_t*>(m_bmem.alloc(nq));
    // decode side: step k's compacted indexes live in their own region [count, int32 | 12 B pad | indexes], so that
    // ONE device->host copy brings the count and (nearly always) all the indexes of a step
    m_idx_region = (16 + nq + 15) / 16 * 16;
    m_CIDX = static_cast<uint8_t*>(m_bmem.alloc(4 * m_idx_region));
    m_DECODED = static_cast<int8_t*>(m_bmem.alloc(4 * nq));
    m_CNT = static_cast<int32_t*>(m_bmem.alloc(sizeof(int32_t) * symbol_blocks(static_cast<int>(nq))));
    m_TOTALS = static_cast<int32_t*>(m_bmem.alloc(sizeof(int32_t) * 4));
    m_h_totals.reserve(16);
    m_h_sym.reserve(4 * nq);
    m_h_z.reserve(P64 * kChZ + 64);
    m_h_idx.reserve(4 * m_idx_region);
    m_h_dec.reserve(4 * nq);
}

void DmciCodec::select_qp(int qp, hipStream_t st)
{
    copy_qp_rows({{m_cur_q_enc, m_q_enc, kChEncDec}, {m_cur_q_dec, m_q_dec, kChEncDec},
                  {m_cur_q_y_enc, m_q_y_enc, kChY}, {m_cur_q_y_dec, m_q_y_dec, kChY}}, qp, st);
}

// ------------------------------------------------------------------------------------ networks
void DmciCodec::run_encoder(hipStream_t st)
{
    const Geometry& g = m_g;
    // dmci_proxy.cpp:92-105 (the per-channel q_scale_enc multiply is applied to the rounded
    // output of enc_1 inside its last conv's epilogue)
    // consecutive full-width blocks hand dc.0 over: block i's launch also computes dc.0 of block i+1
    bool handed = m_enc1.feeds(m_enc2[0]);
    m_enc1.forward(View(m_U, kChSrc, kChSrc), View(m_F, kChEncDec, kChEncDec), g.H8, g.W8, m_s, st,
                   false, nullptr, m_cur_q_enc, View(), handed ? &m_enc2[0] : nullptr);
    const View f(m_F, kChEncDec, kChEncDec);
    for (int i = 0; i < 6; ++i) {
        const DcbW* next = (i < 5 && m_enc2[i].feeds(m_enc2[i + 1])) ? &m_enc2[i + 1] : nullptr;
        m_enc2[i].forward(f, f, g.H8, g.W8, m_s, st, false, nullptr, nullptr, View(), next, handed);
        handed = next != nullptr;
    }
    ConvKxKDesc d;
    d.x = m_F; d.ldx = kChEncDec; d.w = m_enc_down.w; d.bias = m_enc_down.b; d.zeros = m_zeros;
    d.y = m_Y; d.ldy = kChY; d.in_h = g.H8; d.in_w = g.W8; d.cin = kChEncDec; d.cout = kChY;
    d.ksize = 3; d.stride = 2; d.pad = 1;
    conv_kxk(d, st);
}

void DmciCodec::run_hyper_and_priors_enc(hipStream_t st)
{
    const Geometry& g = m_g;
    if (g.padded()) {
        replicate_pad(m_Y, kChY, g.H16, g.W16, kChY, g.H16p - g.H16, g.W16p - g.W16, m_Ypad, kChY, st);
    }
    m_henc0.forward(View(m_Ypad, kChY, kChY), View(m_Z1, kChZ, kChZ), g.H16p, g.W16p, m_s, st);
    m_henc1.forward(View(m_Z1, kChZ, kChZ), View(m_Z2a, kChZ, kChZ), View(m_Z2, kChZ, kChZ), g.H16p, g.W16p,
                    m_zeros, m_s, st);
    m_henc2.forward(View(m_Z2, kChZ, kChZ), View(m_Z3a, kChZ, kChZ), View(m_Z3, kChZ, kChZ), g.H32, g.W32,
                    m_zeros, m_s, st);
    round_z(m_Z3, m_ZH, m_ZI8, g.P64() * kChZ, st);
    run_priors_from_zhat(st);
}

void DmciCodec::run_priors_from_zhat(hipStream_t st)
{
    const Geometry& g = m_g;
    m_hdec0.forward(View(m_ZH, kChZ, kChZ), View(m_H1a, kChZ, kChZ), View(m_H1, kChZ, kChZ), g.H64, g.W64, m_s, st);
    m_hdec1.forward(View(m_H1, kChZ, kChZ), View(m_H2a, kChZ, kChZ), View(m_H2, kChZ, kChZ), g.H32, g.W32, m_s, st);
    m_hdec2.forward(View(m_H2, kChZ, kChZ), View(m_HP, kChY, kChY), g.H16p, g.W16p, m_s, st);
    const View pf(m_PF, 2 * kChY, 2 * kChY);
    {   // a chain of full-width blocks: each launch also computes dc.0 of the block behind it (DcbW::feeds)
        const DcbW* n1 = m_fus[0].feeds(m_fus[1]) ? &m_fus[1] : nullptr;
        const DcbW* n2 = m_fus[1].feeds(m_fus[2]) ? &m_fus[2] : nullptr;
        m_fus[0].forward(View(m_HP, kChY, kChY), pf, g.H16p, g.W16p, m_s, st, false, nullptr, nullptr, View(), n1, false);
        m_fus[1].forward(pf, pf, g.H16p, g.W16p, m_s, st, false, nullptr, nullptr, View(), n2, n1 != nullptr);
        // ... and the last one the conv that closes the chain (y_prior_fusion.conv.3 -> params)
        const FinCall fin(m_fus3, m_PARAMSp, 2 * kChY);
        m_fus[2].forward(pf, pf, g.H16p, g.W16p, m_s, st, false, nullptr, nullptr, View(), nullptr, n2 != nullptr, &fin);
    }
    if (g.padded()) {
        crop(m_PARAMSp, 2 * kChY, g.W16p, m_PARAMS, 2 * kChY, g.H16, g.W16, 2 * kChY, st);
    }
    {   // y_spatial_prior_reduction -> second half of the adaptor input (free torch.cat)
        Conv1x1Desc d;
        d.x = m_PARAMS; d.ldx = 2 * kChY; d.w = m_reduction.w; d.bias = m_reduction.b;
        d.y = m_CAT + kChY; d.ldy = 2 * kChY; d.pixels = g.P16(); d.cin = 2 * kChY; d.cout = kChY;
        conv1x1(d, st);
    }
}

void DmciCodec::run_spatial_prior(int k, hipStream_t st)
{
    const Geometry& g = m_g;
    const View ad(m_AD, 2 * kChY, 2 * kChY);
    const DcbW* next = m_sp_adaptor[k].feeds(m_sp[0]) ? &m_sp[0] : nullptr;
    m_sp_adaptor[k].forward(View(m_CAT, 2 * kChY, 2 * kChY), ad, g.H16, g.W16, m_s, st, false, nullptr, nullptr, View(), next, false);
    for (int i = 0; i < 3; ++i) {
        const bool handed = next != nullptr;
        next = (i < 2 && m_sp[i].feeds(m_sp[i + 1])) ? &m_sp[i + 1] : nullptr;
        const FinCall fin(m_sp3, m_SP, 2 * kChY);          // y_spatial_prior.conv.3 closes the chain
        m_sp[i].forward(ad, ad, g.H16, g.W16, m_s, st, false, nullptr, nullptr, View(), next, handed, i == 2 ? &fin : nullptr);
    }
}

void DmciCodec::run_decoder(half_t* x_hat, hipStream_t st)
{
    const Geometry& g = m_g;
    // dmci_proxy.cpp:14-33
    bool handed = m_dec_up.block.feeds(m_dec1[0]);
    m_dec_up.forward(View(m_YHAT, kChY, kChY), View(m_D0, kChEncDec, kChEncDec), View(m_D1, kChEncDec, kChEncDec),
                     g.H16, g.W16, m_s, st, nullptr, nullptr, handed ? &m_dec1[0] : nullptr);
    const View d1(m_D1, kChEncDec, kChEncDec);
    for (int i = 0; i < 12; ++i) {
        const DcbW* next = (i < 11 && m_dec1[i].feeds(m_dec1[i + 1])) ? &m_dec1[i + 1] : nullptr;
        m_dec1[i].forward(d1, d1, g.H8, g.W8, m_s, st, false, nullptr, i == 11 ? m_cur_q_dec : nullptr, View(),
                          next, handed);
        handed = next != nullptr;
    }
    m_dec2.forward(d1, View(m_R, kChSrc, kChSrc), g.H8, g.W8, m_s, st);
    shuffle8(m_R, kChSrc, g.H8, g.W8, 3, true, x_hat, st);
}

void DmciCodec::enc_stage0(hipStream_t st)
{
    const Geometry& g = m_g;
    run_encoder(st);
    run_hyper_and_priors_enc(st);
    mul_channel(m_Y, kChY, m_cur_q_y_enc, m_Y, kChY, g.P16(), kChY, st);
    const int nq = g.P16() * (kChY / 4);
    for (int k = 0; k < 4; ++k) {
        const half_t* prm = k == 0 ? m_PARAMS : m_SP;
        YStepEnc d;
        d.y = m_Y; d.ldy = kChY;
        d.scales = prm; d.lds = 2 * kChY;
        d.means = prm + kChY; d.ldm = 2 * kChY;
        d.y_hat_acc = m_CAT; d.ldacc = 2 * kChY;
        d.sym = m_SYM; d.cond = m_COND; d.block_count = m_CNT;
        d.H = g.H16; d.W = g.W16; d.C = kChY; d.step = k; d.skip_thres = m_skip_thres; d.first = (k == 0);
        y_step_enc(d, st);
        compact(m_SYM, 2, m_COND, m_CNT, nq, m_COMP, m_TOTALS, k, st);
        if (k < 3) run_spatial_prior(k, st);
    }
    // (y_hat_so_far + y_hat_3) * q_scale_y_dec  (add_and_multiply_broadcast, stream.cu:8-38)
    mul_channel(m_CAT, 2 * kChY, m_cur_q_y_dec, m_YHAT, kChY, g.P16(), kChY, st);
}

// ------------------------------------------------------------------------------------ compress
int DmciCodec::compress(const half_t* x, int height, int width, int qp, half_t* x_hat, hipStream_t user)
{
    prepare(height, width);
    hipStream_t st = enter(user);
    select_qp(qp, st);
    pad_unshuffle8(x, height, width, 3, m_U, m_g.H8, m_g.W8, st);   // outside the graph: x varies
    run_stage(kEnc0, st, [&] { enc_stage0(st); });
    submit(st, [this, qp] { entropy_encode(qp); });
    // the reconstruction runs on the GPU while the worker thread entropy-codes on the host
    bind_stage_arg(kEnc1, x_hat);
    run_stage(kEnc1, st, [&] { run_decoder(x_hat, st); });
    leave(user);
    wait_job();
    return m_ec_parallel;
}

void DmciCodec::entropy_encode(int qp)
{
    // dmci_proxy.cpp:809-845: wait for the symbols, copy them out, code groups 3,2,1,0 then z
    const Geometry& g = m_g;
    hip_check(hipMemcpyAsync(m_h_totals.get(), m_TOTALS, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, m_io_stream), "D2H totals");
    const int nz = g.P64() * kChZ;
    hip_check(hipMemcpyAsync(m_h_z.get(), m_ZI8, nz, hipMemcpyDeviceToHost, m_io_stream), "D2H z");
    hip_check(hipStreamSynchronize(m_io_stream), "sync io");
    int base[4], total = 0;
    for (int k = 0; k < 4; ++k) {
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
    for (int k = 3; k >= 0; --k) m_enc.push_y(m_h_sym.get() + base[k], m_h_totals[k]);
    m_enc.push_z(m_h_z.get(), nz, qp * kChZ, kChZ);
    m_enc.flush();
}

// ------------------------------------------------------------------------------------ decompress
void DmciCodec::decompress(const uint8_t* bits, size_t nbytes, int qp, int height, int width,
                           int ec_parallel, half_t* x_hat, hipStream_t user)
{
    prepare(height, width);
    const Geometry& g = m_g;
    hipStream_t st = enter(user);
    select_qp(qp, st);
    m_dec.set_parallel(ec_parallel);
    m_dec.set_stream(bits, nbytes);
    const int nz = g.P64() * kChZ;
    const int nq = g.P16() * (kChY / 4);
    // DCVC_TIMING: where a decompress() call spends its host time (f3: the entropy decoder's share)
    static const bool trace = getenv("DCVC_TIMING") != nullptr;
    using clk = std::chrono::steady_clock;
    auto us_since = [](clk::time_point t) { return std::chrono::duration<double, std::micro>(clk::now() - t).count(); };
    const auto t_call = clk::now();
    double us_rans = 0, us_wait = 0;
    long n_sym = 0;
    auto t_z = clk::now();
    m_dec.decode_z(nz, qp * kChZ, kChZ, m_h_z.get());
    us_rans += us_since(t_z);
    hip_check(hipMemcpyAsync(m_ZI8, m_h_z.get(), nz, hipMemcpyHostToDevice, st), "H2D z");

    auto index_step = [&](int k) {
        YStepDecIndex d;
        d.scales = k == 0 ? m_PARAMS : m_SP; d.lds = 2 * kChY;
        d.index = m_IDX; d.cond = m_COND; d.block_count = m_CNT;
        d.H = g.H16; d.W = g.W16; d.C = kChY; d.step = k; d.skip_thres = m_skip_thres;
        y_step_dec_index(d, st);
        uint8_t* region = m_CIDX + static_cast<size_t>(k) * m_idx_region;
        compact(m_IDX, 1, m_COND, m_CNT, nq, region + 16, reinterpret_cast<int32_t*>(region), 0, st);
    };
    run_stage(kDec0, st, [&] {
        int8_to_half(m_ZI8, m_ZH, nz, st);
        run_priors_from_zhat(st);
        index_step(0);
    });
    bind_stage_arg(kDec1 + 3, x_hat);
    // the first copy of a step takes the count and up to kFirst index bytes (a 1080p step has 50-130 k of its 522 k
    // possible symbols): one copy + one synchronisation per step instead of two of each (the second pair cost
    // ~25 us of GPU idle time per step in the kernel trace)
    static const size_t kFirst = [] { const char* e = getenv("DCVC_IDX_FIRST_KB"); return static_cast<size_t>(e ? atoi(e) : 192) * 1024; }();
    for (int k = 0; k < 4; ++k) {
        // one GPU -> CPU -> GPU round trip per autoregressive step (dmci_proxy.cpp:857-871)
        auto t_w = clk::now();
        const uint8_t* region = m_CIDX + static_cast<size_t>(k) * m_idx_region;
        uint8_t* h_region = m_h_idx.get() + static_cast<size_t>(k) * m_idx_region;
        const size_t first = std::min(m_idx_region, 16 + kFirst);
        hip_check(hipMemcpyAsync(h_region, region, first, hipMemcpyDeviceToHost, st), "D2H count + indexes");
        hip_check(hipStreamSynchronize(st), "sync");
        const int n = *reinterpret_cast<const int32_t*>(h_region);
        if (n < 0 || static_cast<size_t>(n) > nq) throw std::runtime_error("DMCI decompress: bad symbol count");
        if (16 + static_cast<size_t>(n) > first) {
            hip_check(hipMemcpyAsync(h_region + first, region + first, 16 + n - first, hipMemcpyDeviceToHost, st), "D2H indexes");
            hip_check(hipStreamSynchronize(st), "sync");
        }
        us_wait += us_since(t_w);
        int8_t* h_dec = m_h_dec.get() + static_cast<size_t>(k) * nq;
        if (n > 0) {
            auto t_r = clk::now();
            m_dec.decode_y(h_region + 16, n, h_dec);
            us_rans += us_since(t_r);
            n_sym += n;
            hip_check(hipMemcpyAsync(m_DECODED + static_cast<size_t>(k) * nq, h_dec, n, hipMemcpyHostToDevice, st), "H2D symbols");
        }
        run_stage(kDec1 + k, st, [&] {
            YStepDecRestore d;
            d.decoded = m_DECODED + static_cast<size_t>(k) * nq; d.cond = m_COND; d.block_count = m_CNT;
            d.totals = m_TOTALS; d.slot = 0;                   // the step's own region: no base to add up
            d.means = (k == 0 ? m_PARAMS : m_SP) + kChY; d.ldm = 2 * kChY;
            d.y_hat_acc = m_CAT; d.ldacc = 2 * kChY;
            d.H = g.H16; d.W = g.W16; d.C = kChY; d.step = k; d.first = (k == 0);
            y_step_dec_restore(d, st);
            if (k < 3) {
                run_spatial_prior(k, st);
                index_step(k + 1);
            } else {
                mul_channel(m_CAT, 2 * kChY, m_cur_q_y_dec, m_YHAT, kChY, g.P16(), kChY, st);
                run_decoder(x_hat, st);
            }
        });
    }
    if (trace) {
        fprintf(stderr, "[dcvc] decompress host %.0f us: entropy decoding %.0f us (%ld y symbols, %zu bytes, %d streams), "
                        "waiting for the GPU %.0f us\n", us_since(t_call), us_rans, n_sym, nbytes, ec_parallel, us_wait);
    }
    leave(user);
}

// ------------------------------------------------------------------------------------ debug
size_t DmciCodec::debug_read(const std::string& name, void* dst, size_t cap, hipStream_t st)
{
    const Geometry& g = m_g;
    const void* src = nullptr;
    size_t bytes = 0;
    if (name == "y") { src = m_Y; bytes = static_cast<size_t>(g.P16()) * kChY * 2; }
    else if (name == "y_hat") { src = m_YHAT; bytes = static_cast<size_t>(g.P16()) * kChY * 2; }
    else if (name == "z_i8") { src = m_ZI8; bytes = static_cast<size_t>(g.P64()) * kChZ; }
    else if (name == "params") { src = m_PARAMS; bytes = static_cast<size_t>(g.P16()) * 2 * kChY * 2; }
    else if (name == "unshuffled") { src = m_U; bytes = static_cast<size_t>(g.P8()) * kChSrc * 2; }
    else if (name == "features") { src = m_F; bytes = static_cast<size_t>(g.P8()) * kChEncDec * 2; }
    else if (name == "totals") { src = m_TOTALS; bytes = 16; }
    else if (name == "symbols") { src = m_COMP; bytes = static_cast<size_t>(g.P16()) * kChY * 2; }
    else throw std::invalid_argument("unknown debug tensor '" + name + "'");
    if (dst != nullptr) {
        hip_check(hipStreamSynchronize(st), "sync");
        hip_check(hipStreamSynchronize(m_cs), "sync");
        hip_check(hipMemcpy(dst, src, std::min(bytes, cap), hipMemcpyDeviceToHost), "debug D2H");
    }
    return bytes;
}

}  // namespace dcvc
