#include "codec/modules.h"

#include <cmath>
#include <cstring>

namespace dcvc {

// ---------------------------------------------------------------- ParamStore
int64_t HostTensor::numel() const
{
    int64_t n = 1;
    for (int64_t d : shape) n *= d;
    return n;
}

void ParamStore::add(const std::string& name, const void* data, int dtype, const int64_t* dims, int ndim)
{
    HostTensor t;
    t.shape.assign(dims, dims + ndim);
    const int64_t n = t.numel();
    if (dtype == 0) {
        t.h.resize(n);
        std::memcpy(t.h.data(), data, n * sizeof(half_t));
    } else if (dtype == 1) {
        t.h.resize(n);
        const float* f = static_cast<const float*>(data);
        for (int64_t i = 0; i < n; ++i) t.h[i] = static_cast<half_t>(f[i]);
    } else if (dtype == 2) {
        t.i.resize(n);
        std::memcpy(t.i.data(), data, n * sizeof(int32_t));
    } else {
        throw std::invalid_argument("set_param: unsupported dtype code for " + name);
    }
    m_map[name] = std::move(t);
}

const HostTensor& ParamStore::at(const std::string& name) const
{
    auto it = m_map.find(name);
    if (it == m_map.end()) {
        throw std::out_of_range("state_dict has no entry '" + name + "'");
    }
    return it->second;
}

// ---------------------------------------------------------------- DeviceArena
DeviceArena::~DeviceArena()
{
    release();
}

void* DeviceArena::alloc(size_t bytes)
{
    void* p = nullptr;
    const size_t sz = (bytes + 255) / 256 * 256 + 256;
    hip_check(hipMalloc(&p, sz), "hipMalloc");
    // the codec streams are non-blocking: they do not order themselves behind the null stream
    // the fill runs on, so the fill has to be finished before the buffer is handed out
    hip_check(hipMemset(p, 0, sz), "hipMemset");
    hip_check(hipStreamSynchronize(nullptr), "hipStreamSynchronize(fill)");
    m_ptrs.push_back(p);
    m_total += sz;
    return p;
}

half_t* DeviceArena::upload(const std::vector<half_t>& host)
{
    half_t* d = alloc_half(host.size());
    hip_check(hipMemcpy(d, host.data(), host.size() * sizeof(half_t), hipMemcpyHostToDevice), "upload");
    return d;
}

void DeviceArena::release()
{
    for (void* p : m_ptrs) (void)hipFree(p);
    m_ptrs.clear();
    m_total = 0;
}

// ---------------------------------------------------------------- weights
namespace {

void expect_conv(const HostTensor& w, int k, const std::string& name)
{
    if (w.shape.size() != 4 || w.shape[2] != k || w.shape[3] != k) {
        throw std::invalid_argument("unexpected weight shape for " + name);
    }
}

}  // namespace

void Conv1x1W::load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix)
{
    const HostTensor& wt = ps.at(prefix + "weight");
    expect_conv(wt, 1, prefix + "weight");
    cout = static_cast<int>(wt.shape[0]);
    cin = static_cast<int>(wt.shape[1]);
    w = mem.upload(wt.h);
    b = ps.has(prefix + "bias") ? mem.upload(ps.at(prefix + "bias").h) : nullptr;
}

void ConvKW::load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix)
{
    const HostTensor& wt = ps.at(prefix + "weight");     // [cout][cin][k][k]
    if (wt.shape.size() != 4 || wt.shape[2] != wt.shape[3]) {
        throw std::invalid_argument("unexpected weight shape for " + prefix + "weight");
    }
    cout = static_cast<int>(wt.shape[0]);
    cin = static_cast<int>(wt.shape[1]);
    k = static_cast<int>(wt.shape[2]);
    std::vector<half_t> r(wt.h.size());
    for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < k * k; ++t)
                r[(static_cast<size_t>(n) * k * k + t) * cin + c] = wt.h[(static_cast<size_t>(n) * cin + c) * k * k + t];
    w = mem.upload(r);
    b = ps.has(prefix + "bias") ? mem.upload(ps.at(prefix + "bias").h) : nullptr;
}

void FinW::load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix)
{
    conv.load(ps, mem, prefix);
    packed = nullptr;
    if (conv.b != nullptr && conv.cin % 128 == 0 && conv.cout % 32 == 0 && conv.cout >= 128 && dcb_nsplit_waves() == 8) {
        packed = mem.alloc_half(dcb_nsplit_fin_halves(conv.cin, conv.cout));
        dcb_nsplit_pack_fin(conv.w, conv.cin, conv.cout, packed, nullptr);
        hip_check(hipStreamSynchronize(nullptr), "hipStreamSynchronize(pack)");
    }
}

namespace {

void run_fin(const FinCall& f, View x, int pixels, hipStream_t st)
{
    Conv1x1Desc d;
    d.x = x.p; d.ldx = x.ld; d.w = f.w->conv.w; d.bias = f.w->conv.b; d.q = f.q;
    d.y = f.y; d.ldy = f.ldy; d.pixels = pixels; d.cin = f.w->conv.cin; d.cout = f.w->conv.cout;
    conv1x1(d, st);
}

}  // namespace

void DcbW::load(const ParamStore& ps, DeviceArena& mem, const std::string& p)
{
    has_adaptor = ps.has(p + "adaptor.weight");
    if (has_adaptor) adaptor.load(ps, mem, p + "adaptor.");
    dc0.load(ps, mem, p + "dc.0.");
    ffn0.load(ps, mem, p + "ffn.0.");
    ffn2.load(ps, mem, p + "ffn.2.");
    c = dc0.cin;
    cdc = dc0.cout;
    cffn = ffn0.cout / 4;
    // depthwise weight [cdc][1][3][3] -> tap major [9][cdc]
    const HostTensor& dwt = ps.at(p + "dc.2.weight");
    if (dwt.shape.size() != 4 || dwt.shape[0] != cdc || dwt.shape[1] != 1 || dwt.shape[2] != 3) {
        throw std::invalid_argument("unexpected depthwise weight shape for " + p + "dc.2.weight");
    }
    std::vector<half_t> r(static_cast<size_t>(9) * cdc);
    for (int ch = 0; ch < cdc; ++ch)
        for (int t = 0; t < 9; ++t) r[static_cast<size_t>(t) * cdc + ch] = dwt.h[static_cast<size_t>(ch) * 9 + t];
    dw = mem.upload(r);
    // dc.3 with the depthwise bias folded through it (layers_proxy.cpp:175-178):
    //   b' = fp16( fp16( sum_c W3[n][c] * b2[c] ) + b3[n] ), fp32 fmaf chain over c
    const HostTensor& w3 = ps.at(p + "dc.3.weight");
    expect_conv(w3, 1, p + "dc.3.weight");
    const HostTensor& b2 = ps.at(p + "dc.2.bias");
    const HostTensor& b3 = ps.at(p + "dc.3.bias");
    dc3.cout = static_cast<int>(w3.shape[0]);
    dc3.cin = static_cast<int>(w3.shape[1]);
    dc3.w = mem.upload(w3.h);
    std::vector<half_t> folded(dc3.cout);
    for (int n = 0; n < dc3.cout; ++n) {
        float acc = 0.f;
        for (int ch = 0; ch < dc3.cin; ++ch) {
            acc = std::fmaf(static_cast<float>(w3.h[static_cast<size_t>(n) * dc3.cin + ch]),
                            static_cast<float>(b2.h[ch]), acc);
        }
        const half_t t = static_cast<half_t>(acc);
        folded[n] = static_cast<half_t>(static_cast<float>(t) + static_cast<float>(b3.h[n]));
    }
    dc3.b = mem.upload(folded);
    if (dc3.cin != cdc || dc3.cout != c || ffn0.cin != c || ffn2.cin != cffn || ffn2.cout != c) {
        throw std::invalid_argument("inconsistent DepthConvBlock shapes under " + p);
    }
    if (dcb_nsplit_supported(c, cdc, cffn) && dc0.b && dc3.b && ffn0.b && ffn2.b) {
        // kernels/dcb_nsplit.hip: per-wave linear streams of MFMA weight fragments, packed once on the device
        packed_main = mem.alloc_half(dcb_nsplit_main_halves(c, cdc));
        packed_dc0 = mem.alloc_half(dcb_nsplit_dc0_halves(c, cdc));
        dcb_nsplit_pack_main(dc3.w, ffn0.w, ffn2.w, c, cdc, packed_main, nullptr);
        dcb_nsplit_pack_dc0(dc0.w, c, cdc, packed_dc0, nullptr);
        if (has_adaptor && adaptor.b != nullptr && dcb_pair_supported(adaptor.cin, c, cdc)) {
            packed_adaptor = mem.alloc_half(dcb_pair_adaptor_halves(adaptor.cin, c));
            dcb_pair_pack_adaptor(adaptor.w, adaptor.cin, c, packed_adaptor, nullptr);
        }
        hip_check(hipStreamSynchronize(nullptr), "hipStreamSynchronize(pack)");
    }
}

bool DcbW::core_fused() const
{
    return nsplit();
}

bool DcbW::one_launch(int H, int W) const
{
    return !nsplit() && !has_adaptor && dcb_tail_supported(H, W, c, cdc, cffn) && dcb_tail_takes_dc0() && ffn0.b != nullptr &&
           ffn2.b != nullptr && dc3.b != nullptr && dc0.b != nullptr;
}

bool DcbW::feeds(const DcbW& next) const
{
    return core_fused() && next.core_fused() && !next.has_adaptor && next.c == c && next.cdc == cdc && nsplit() == next.nsplit();
}

void DcbW::forward(View x, View y, int H, int W, const Scratch& s, hipStream_t st, bool shortcut,
                   const half_t* q_fused, const half_t* q_after, View alt, const DcbW* next, bool dc0_done, const FinCall* fin) const
{
    const int P = H * W;
    if (fin != nullptr && (next != nullptr || fin->w == nullptr || fin->w->conv.cin != c)) {
        throw std::invalid_argument("DepthConvBlock: a closing conv sits behind the LAST block of a chain and reads its output");
    }
    if ((next != nullptr && !feeds(*next)) || (dc0_done && (!core_fused() || has_adaptor))) {
        throw std::invalid_argument("DepthConvBlock: dc.0 hand-over between blocks that do not support it");
    }
    if (static_cast<size_t>(P) * cdc > s.elems || static_cast<size_t>(P) * cffn > s.elems) {
        throw std::runtime_error("DepthConvBlock: scratch planes too small");
    }
    View in = x;
    // a block that computes its own dc.0 starts from the canonical planes: which plane a launch reads and writes is then a function of
    // the chain's structure alone (launch sequences captured into graphs at different times agree with each other)
    if (!dc0_done) s.hand = 0;
    half_t* const p1 = s.hand != 0 ? s.t2 : s.t1;      // dc.0's output (ours, or handed over by the previous launch)
    half_t* const p2 = s.hand != 0 ? s.t1 : s.t2;      // the depthwise conv's output / the next block's dc.0 output
    if (has_adaptor && shortcut) {
        // the adaptor output would have to survive the in-place dc.3 / ffn.2 updates
        throw std::invalid_argument("DepthConvBlock with adaptor and shortcut is not a DCVC-UF block");
    }
    const bool tail = !nsplit() && dcb_tail_supported(H, W, c, cdc, cffn) && ffn0.b != nullptr && ffn2.b != nullptr &&
                      dc3.b != nullptr && dc0.b != nullptr;
    if (has_adaptor) {
        // the one-launch block (dcb_tail with dc.0 inside) reads its input's neighbours, so the adaptor output must not be
        // the block output: without a spare buffer from the caller it goes to the third scratch plane, which that launch leaves alone
        if (alt.p == nullptr && tail && static_cast<size_t>(P) * c <= s.elems) alt = View(s.t3, c, c);
        const View a = (alt.p != nullptr && alt.p != y.p) ? View(alt.p, alt.ld, c) : y;
        if (packed_adaptor != nullptr && nsplit() && a.p != x.p) {
            // adaptor + dc.0 in ONE launch (kernels/dcb_pair8_kernel.h): the adaptor output stays in LDS as dc.0's operand
            DcbPairDesc d;
            d.x = x.p; d.ldx = x.ld; d.wa = packed_adaptor; d.ba = adaptor.b; d.w1 = packed_dc0; d.b1 = dc0.b;
            d.y = a.p; d.ldy = a.ld; d.t1 = p1; d.ldt1 = cdc; d.pixels = P; d.cin = adaptor.cin; d.c = c; d.ci = cdc;
            dcb_pair(d, st);
            dc0_done = true;
        } else {
            Conv1x1Desc d;
            d.x = x.p; d.ldx = x.ld; d.w = adaptor.w; d.bias = adaptor.b;
            d.y = a.p; d.ldy = a.ld; d.pixels = P; d.cin = adaptor.cin; d.cout = adaptor.cout;
            conv1x1(d, st);
        }
        in = a;
    } else if (shortcut && x.p == y.p) {
        throw std::invalid_argument("DepthConvBlock with shortcut cannot run in place");
    }
    const bool tail_dc0 = tail && dcb_tail_takes_dc0() && in.p != y.p;      // reads neighbours' input: not in place
    if (!tail_dc0 && !dc0_done) {   // dc.0 + WSiLU
        Conv1x1Desc d;
        d.x = in.p; d.ldx = in.ld; d.w = dc0.w; d.bias = dc0.b; d.wsilu = true;
        d.y = p1; d.ldy = cdc; d.pixels = P; d.cin = c; d.cout = cdc;
        conv1x1(d, st);
    }
    if (tail) {
        // [dc.0 +] depthwise + dc.3 + ffn.0 + ffn.2 in one launch (kernels/dcb_tail.hip)
        DcbTailDesc d;
        if (tail_dc0) { d.w1 = dc0.w; d.b1 = dc0.b; }
        d.t = p1; d.ldt = cdc; d.dw = dw; d.x = in.p; d.ldx = in.ld;
        d.w3 = dc3.w; d.b3 = dc3.b; d.w0 = ffn0.w; d.b0 = ffn0.b; d.w2 = ffn2.w; d.b2 = ffn2.b;
        d.q = q_fused; d.q2 = q_after; d.y = y.p; d.ldy = y.ld;
        d.H = H; d.W = W; d.c = c; d.cdc = cdc; d.cffn = cffn; d.shortcut = shortcut;
        dcb_tail(d, st);
        if (fin != nullptr) run_fin(*fin, y, P, st);
        return;
    }
    // the narrow blocks take their depthwise conv into the block launch (dcb_nsplit8_kernel.h, DW): dc.0's output is read from one
    // scratch plane, the next block's written to the other
    // (the kernel addresses dc.0's output with 32-bit byte offsets from its base)
    const bool dw_inside = nsplit() && dcb_nsplit_dw_supported(c, cdc, P) && static_cast<long long>(P) * cdc * 2 < (1LL << 31);
    if (!dw_inside) dwconv3x3(p1, cdc, dw, p2, cdc, H, W, cdc, st);
    if (nsplit()) {
        // [depthwise +] dc.3 + ffn.0 + ffn.2 (+ the next block's dc.0) in one launch, activations in LDS, weights per wave from L2
        DcbNsplitDesc d;
        d.ldt = cdc; d.x = in.p; d.ldx = in.ld;
        if (dw_inside) { d.t1 = p1; d.wdw = dw; d.width = W; }
        else d.t2 = p2;
        d.wmain = packed_main; d.b3 = dc3.b; d.b0 = ffn0.b; d.b2 = ffn2.b;
        d.q = q_fused; d.q2 = q_after; d.y = y.p; d.ldy = y.ld; d.pixels = P; d.c = c; d.ci = cdc; d.shortcut = shortcut;
        if (next != nullptr) {
            d.wnext = next->packed_dc0; d.b1n = next->dc0.b; d.t1n = dw_inside ? p2 : p1; d.ldt1 = next->cdc;
            if (dw_inside) s.hand ^= 1;
        }
        const bool fin_inside = fin != nullptr && fin->w->packed != nullptr && dcb_nsplit_fin_supported(c, cdc, fin->w->conv.cout);
        if (fin_inside) {
            d.wfin = fin->w->packed; d.bfin = fin->w->conv.b; d.qfin = fin->q; d.yfin = fin->y; d.ldyfin = fin->ldy;
            d.nfin = fin->w->conv.cout;
            if (!fin->keep_block_output && !shortcut) d.y = nullptr;      // (with the block shortcut y may alias x: keep it simple)
        }
        dcb_nsplit(d, st);
        if (fin != nullptr && !fin_inside) run_fin(*fin, y, P, st);
        return;
    }
    {   // dc.3 (+ folded depthwise bias) + shortcut
        Conv1x1Desc d;
        d.x = p2; d.ldx = cdc; d.w = dc3.w; d.bias = dc3.b; d.r1 = in.p; d.ldr1 = in.ld;
        d.y = y.p; d.ldy = y.ld; d.pixels = P; d.cin = cdc; d.cout = c;
        conv1x1(d, st);
    }
    if (ffn_fused_supported(P, c, cffn) && ffn0.b != nullptr && ffn2.b != nullptr) {
        // ffn.0 + ffn.2 in one launch: the chunk-added tensor stays in LDS (kernels/ffn_fused.hip)
        FfnFusedDesc d;
        d.x = y.p; d.ldx = y.ld; d.w0 = ffn0.w; d.b0 = ffn0.b; d.w2 = ffn2.w; d.b2 = ffn2.b;
        if (shortcut) { d.r2 = in.p; d.ldr2 = in.ld; }
        d.q = q_fused; d.q2 = q_after;
        d.y = y.p; d.ldy = y.ld; d.pixels = P; d.c = c; d.cffn = cffn;
        ffn_fused(d, st);
        if (fin != nullptr) run_fin(*fin, y, P, st);
        return;
    }
    {   // ffn.0 + WSiLU + chunk-add: the 4x expanded tensor never reaches HBM
        Conv1x1Desc d;
        d.x = y.p; d.ldx = y.ld; d.w = ffn0.w; d.bias = ffn0.b; d.wsilu = true; d.chunk_add = true;
        d.y = s.t3; d.ldy = cffn; d.pixels = P; d.cin = c; d.cout = ffn0.cout;
        conv1x1(d, st);
    }
    {   // ffn.2 + shortcut (+ block input when `shortcut`) (* quant)
        Conv1x1Desc d;
        d.x = s.t3; d.ldx = cffn; d.w = ffn2.w; d.bias = ffn2.b; d.r1 = y.p; d.ldr1 = y.ld;
        if (shortcut) { d.r2 = in.p; d.ldr2 = in.ld; }
        d.q = q_fused; d.q2 = q_after;
        d.y = y.p; d.ldy = y.ld; d.pixels = P; d.cin = cffn; d.cout = c;
        conv1x1(d, st);
    }
    if (fin != nullptr) run_fin(*fin, y, P, st);
}

void Stride2W::load(const ParamStore& ps, DeviceArena& mem, const std::string& p, bool with_shortcut)
{
    shortcut = with_shortcut;
    const HostTensor& wt = ps.at(p + "down.weight");     // [cout][4*cin][1][1], channel = c*4 + dy*2 + dx
    expect_conv(wt, 1, p + "down.weight");
    cout = static_cast<int>(wt.shape[0]);
    cin = static_cast<int>(wt.shape[1]) / 4;
    std::vector<half_t> r(wt.h.size());
    for (int n = 0; n < cout; ++n)
        for (int ch = 0; ch < cin; ++ch)
            for (int t = 0; t < 4; ++t)
                r[(static_cast<size_t>(n) * 4 + t) * cin + ch] = wt.h[static_cast<size_t>(n) * 4 * cin + ch * 4 + t];
    w = mem.upload(r);
    b = mem.upload(ps.at(p + "down.bias").h);
    block.load(ps, mem, p + "conv.");
}

void Stride2W::forward(View x, View tmp, View y, int H, int W, const half_t* zeros, const Scratch& s,
                       hipStream_t st) const
{
    ConvKxKDesc d;
    d.x = x.p; d.ldx = x.ld; d.w = w; d.bias = b; d.zeros = zeros;
    d.y = tmp.p; d.ldy = tmp.ld; d.in_h = H; d.in_w = W; d.cin = cin; d.cout = cout;
    d.ksize = 2; d.stride = 2; d.pad = 0;
    // a caller without a buffer to spare passes tmp = y: the block would run in place, and the one-launch form (dc.0 on the
    // halo of a patch) cannot. The third scratch plane is free in that form: the conv's output goes there instead
    if (tmp.p == y.p && block.one_launch(H / 2, W / 2) && static_cast<size_t>(H / 2) * (W / 2) * cout <= s.elems) {
        tmp = View(s.t3, cout, cout);
        d.y = tmp.p; d.ldy = tmp.ld;
    }
    conv_kxk(d, st);
    block.forward(tmp, y, H / 2, W / 2, s, st, shortcut);
}

void SubpelW::load(const ParamStore& ps, DeviceArena& mem, const std::string& p)
{
    const HostTensor& wt = ps.at(p + "conv.0.weight");   // [4*cout][cin][k][k], row = co*4 + dy*2 + dx
    if (wt.shape.size() != 4 || wt.shape[2] != wt.shape[3]) {
        throw std::invalid_argument("unexpected weight shape for " + p + "conv.0.weight");
    }
    cout = static_cast<int>(wt.shape[0]) / 4;
    cin = static_cast<int>(wt.shape[1]);
    k = static_cast<int>(wt.shape[2]);
    if (ps.has(p + "conv.0.bias")) {
        ConvKW c;
        c.load(ps, mem, p + "conv.0.");
        w = c.w;
        b = c.b;
        return;
    }
    if (k != 1) throw std::invalid_argument("SubpelConv2x without bias must have kernel 1: " + p);
    b = nullptr;
    std::vector<half_t> r(wt.h.size());
    for (int co = 0; co < cout; ++co)
        for (int t = 0; t < 4; ++t)
            std::memcpy(&r[(static_cast<size_t>(t) * cout + co) * cin], &wt.h[(static_cast<size_t>(co) * 4 + t) * cin],
                        cin * sizeof(half_t));
    w = mem.upload(r);
}

void SubpelW::forward(View x, View y, int H, int W, hipStream_t st, half_t* tmp, const half_t* zeros) const
{
    if (b == nullptr) {
        TConv2x2Desc d;
        d.x = x.p; d.ldx = x.ld; d.w = w; d.y = y.p; d.ldy = y.ld;
        d.in_h = H; d.in_w = W; d.cin = cin; d.cout = cout;
        tconv2x2(d, st);
        return;
    }
    if (tmp == nullptr) throw std::invalid_argument("biased SubpelConv2x needs a temporary");
    if (k == 1) {
        Conv1x1Desc d;
        d.x = x.p; d.ldx = x.ld; d.w = w; d.bias = b; d.y = tmp; d.ldy = 4 * cout;
        d.pixels = H * W; d.cin = cin; d.cout = 4 * cout;
        conv1x1(d, st);
    } else {
        ConvKxKDesc d;
        d.x = x.p; d.ldx = x.ld; d.w = w; d.bias = b; d.zeros = zeros;
        d.y = tmp; d.ldy = 4 * cout; d.in_h = H; d.in_w = W; d.cin = cin; d.cout = 4 * cout;
        d.ksize = k; d.stride = 1; d.pad = k / 2;
        conv_kxk(d, st);
    }
    shuffle2(tmp, 4 * cout, H, W, cout, y.p, y.ld, st);
}

void UpsampleW::load(const ParamStore& ps, DeviceArena& mem, const std::string& p, bool with_shortcut)
{
    shortcut = with_shortcut;
    up.load(ps, mem, p + "up.");
    block.load(ps, mem, p + "conv.");
}

void UpsampleW::forward(View x, View tmp, View y, int H, int W, const Scratch& s, hipStream_t st,
                        half_t* up_tmp, const half_t* zeros, const DcbW* next) const
{
    if (tmp.p == y.p && block.one_launch(2 * H, 2 * W) && static_cast<size_t>(4) * H * W * up.cout <= s.elems) {
        tmp = View(s.t3, up.cout, up.cout);      // (as Stride2W::forward: the one-launch block cannot run in place)
    }
    up.forward(x, tmp, H, W, st, up_tmp, zeros);
    block.forward(tmp, y, 2 * H, 2 * W, s, st, shortcut, nullptr, nullptr, View(), next);
}

void DcbChain::load(const ParamStore& ps, DeviceArena& mem, const std::string& prefix)
{
    blocks.clear();
    for (int i = 0; ps.has(prefix + std::to_string(i) + ".dc.0.weight"); ++i) {
        blocks.emplace_back();
        blocks.back().load(ps, mem, prefix + std::to_string(i) + ".");
    }
    if (blocks.empty()) throw std::invalid_argument("no DepthConvBlocks under " + prefix);
}

void run_dcb_chain(const DcbW* blocks, int n, View x, View tmp, View y, int H, int W, const Scratch& s,
                   hipStream_t st, const half_t* q_fused_last, View tmp2, const FinCall* fin, const DcbW* after, bool first_dc0_done)
{
    if (after != nullptr && (fin != nullptr || !blocks[n - 1].feeds(*after))) {
        throw std::invalid_argument("run_dcb_chain: the chain behind this one cannot take its dc.0 from the last launch");
    }
    auto next_of = [&](int i) -> const DcbW* {
        if (i + 1 < n) return blocks[i].feeds(blocks[i + 1]) ? &blocks[i + 1] : nullptr;
        return after;
    };
    if (tmp2.p == nullptr) {
        View cur = x;
        bool handed = first_dc0_done;   // the previous block left this block's dc.0 output in s.t1
        for (int i = 0; i < n; ++i) {
            const View out = (i == n - 1) ? y : tmp;
            const DcbW* next = next_of(i);
            blocks[i].forward(cur, out, H, W, s, st, false, i == n - 1 ? q_fused_last : nullptr, nullptr, View(),
                              next, handed, i == n - 1 ? fin : nullptr);
            handed = next != nullptr;
            cur = out;
        }
        return;
    }
    // ping-pong: out[n-1] = y, the outputs before it alternate between the two temporaries so that no
    // block runs in place (the one-launch block kernel needs input != output)
    const View a = tmp, b = tmp2;
    View cur = x;
    bool handed = first_dc0_done;
    for (int i = 0; i < n; ++i) {
        View out = y;
        if (i != n - 1) {
            const bool y_is_a = y.p == a.p;
            const int back = n - 1 - i;                         // distance from the last block
            out = (back % 2 == 1) ? (y_is_a ? b : a) : (y_is_a ? a : b);
        }
        const View spare = (out.p == a.p) ? b : a;              // for an adaptor: neither input nor output
        const DcbW* next = next_of(i);
        blocks[i].forward(cur, out, H, W, s, st, false, i == n - 1 ? q_fused_last : nullptr, nullptr,
                          cur.p == spare.p ? View() : spare, next, handed, i == n - 1 ? fin : nullptr);
        handed = next != nullptr;
        cur = out;
    }
}

void DcbChain::forward(View x, View tmp, View y, int H, int W, const Scratch& s, hipStream_t st,
                       const half_t* q_fused_last, View tmp2, const FinCall* fin) const
{
    run_dcb_chain(blocks.data(), static_cast<int>(blocks.size()), x, tmp, y, H, W, s, st, q_fused_last, tmp2, fin);
}

}  // namespace dcvc
