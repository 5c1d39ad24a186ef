// C ABI of the individual kernels (include/dcvc_amd_ops.h).
#include "capi_common.h"
#include <memory>
#include <mutex>
#include <stdexcept>
#include "dcvc_amd_ops.h"
#include "kernels/ops.h"

namespace {

using dcvc::half_t;

inline const half_t* H(const void* p) { return static_cast<const half_t*>(p); }
inline half_t* H(void* p) { return static_cast<half_t*>(p); }
inline hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }

// one lazily allocated page of zeros for padding taps
const half_t* zero_page()
{
    static half_t* z = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        dcvc::hip_check(hipMalloc(&z, 4096), "hipMalloc(zero page)");
        dcvc::hip_check(hipMemset(z, 0, 4096), "hipMemset(zero page)");
    });
    return z;
}

}  // namespace

extern "C" {

int dcvc_conv1x1(const void* x, int ldx, const void* w, const void* bias, const void* r1, int ldr1,
                 const void* r2, int ldr2, const void* q, const void* q2, void* y, int ldy,
                 int pixels, int cin, int cout, int flags, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::Conv1x1Desc d;
        d.x = H(x); d.ldx = ldx; d.w = H(w); d.bias = H(bias);
        d.r1 = H(r1); d.ldr1 = ldr1; d.r2 = H(r2); d.ldr2 = ldr2;
        d.q = H(q); d.q2 = H(q2); d.y = H(y); d.ldy = ldy;
        d.pixels = pixels; d.cin = cin; d.cout = cout;
        dcvc::kernels_init();
        d.wsilu = (flags & DCVC_CONV_WSILU) != 0;
        d.chunk_add = (flags & DCVC_CONV_CHUNK_ADD) != 0;
        dcvc::conv1x1(d, S(stream));
    });
}

int dcvc_conv_kxk(const void* x, int ldx, const void* w, const void* bias, void* y, int ldy,
                  int in_h, int in_w, int cin, int cout, int ksize, int stride, int pad, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::ConvKxKDesc d;
        d.x = H(x); d.ldx = ldx; d.w = H(w); d.bias = H(bias); d.zeros = zero_page();
        d.y = H(y); d.ldy = ldy; d.in_h = in_h; d.in_w = in_w; d.cin = cin; d.cout = cout;
        d.ksize = ksize; d.stride = stride; d.pad = pad;
        dcvc::conv_kxk(d, S(stream));
    });
}

int dcvc_tconv2x2(const void* x, int ldx, const void* w, void* y, int ldy, int in_h, int in_w,
                  int cin, int cout, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::TConv2x2Desc d;
        d.x = H(x); d.ldx = ldx; d.w = H(w); d.y = H(y); d.ldy = ldy;
        d.in_h = in_h; d.in_w = in_w; d.cin = cin; d.cout = cout;
        dcvc::tconv2x2(d, S(stream));
    });
}

int dcvc_dwconv3x3(const void* x, int ldx, const void* w, void* y, int ldy, int Hh, int W, int C,
                   void* stream)
{
    return dcvc::guarded([&] { dcvc::dwconv3x3(H(x), ldx, H(w), H(y), ldy, Hh, W, C, S(stream)); });
}

int dcvc_pad_unshuffle8(const void* x, int Hh, int W, int C3, void* out, int H8, int W8, void* stream)
{
    return dcvc::guarded([&] { dcvc::pad_unshuffle8(H(x), Hh, W, C3, H(out), H8, W8, S(stream)); });
}

int dcvc_shuffle8(const void* in, int ldin, int H8, int W8, int C3, int clamp, void* out, void* stream)
{
    return dcvc::guarded([&] { dcvc::shuffle8(H(in), ldin, H8, W8, C3, clamp != 0, H(out), S(stream)); });
}

int dcvc_shuffle2(const void* in, int ldin, int Hh, int W, int C, void* out, int ldout, void* stream)
{
    return dcvc::guarded([&] { dcvc::shuffle2(H(in), ldin, Hh, W, C, H(out), ldout, S(stream)); });
}

int dcvc_replicate_pad(const void* in, int ldin, int Hh, int W, int C, int pad_b, int pad_r,
                       void* out, int ldout, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::replicate_pad(H(in), ldin, Hh, W, C, pad_b, pad_r, H(out), ldout, S(stream));
    });
}

int dcvc_crop(const void* in, int ldin, int Win, void* out, int ldout, int Hh, int W, int C, void* stream)
{
    return dcvc::guarded([&] { dcvc::crop(H(in), ldin, Win, H(out), ldout, Hh, W, C, S(stream)); });
}

int dcvc_mul_channel(const void* x, int ldx, const void* q, void* y, int ldy, int pixels, int C,
                     void* stream)
{
    return dcvc::guarded([&] { dcvc::mul_channel(H(x), ldx, H(q), H(y), ldy, pixels, C, S(stream)); });
}

int dcvc_ffn_fused(const void* x, int ldx, const void* w0, const void* b0, const void* w2, const void* b2,
                   const void* r2, int ldr2, const void* q, const void* q2, void* y, int ldy,
                   int pixels, int c, int cffn, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::kernels_init();
        dcvc::FfnFusedDesc d;
        d.x = H(x); d.ldx = ldx; d.w0 = H(w0); d.b0 = H(b0); d.w2 = H(w2); d.b2 = H(b2);
        d.r2 = H(r2); d.ldr2 = ldr2; d.q = H(q); d.q2 = H(q2); d.y = H(y); d.ldy = ldy;
        d.pixels = pixels; d.c = c; d.cffn = cffn;
        dcvc::ffn_fused(d, S(stream));
    });
}

namespace {
// The N-split kernel wants its per-wave fragment streams, not the reference's row-major matrices. The codecs pack once
// at set_param time (dcvc::DcbW::load). The operator-level entry points below serve tests and tools:
//   dcvc_dcb_nsplit          packs on EVERY call into stream-ordered temporaries (hipMallocAsync / hipFreeAsync on the
//                            caller's stream): always reads the weights as they are now. (Round 3 cached packed copies per
//                            weight POINTER: a caller that rewrote its weights in place, or whose allocator handed the same
//                            address out again, silently got the old weights - advisor, round 3.)
//   dcvc_dcb_nsplit_pack / _packed / _free   explicit handle for callers that launch many times (tools/probes/core_bench).
// Device buffers of the entry points below, released on every path out (advisor, round 4: a throwing pack launch leaked them).
// Stream-ordered temporaries (hipMallocAsync / hipFreeAsync on the caller's stream) or plain allocations of a device.
struct AsyncBuf {
    void* p = nullptr;
    hipStream_t st = nullptr;
    AsyncBuf(size_t bytes, hipStream_t stream) : st(stream) { dcvc::hip_check(hipMallocAsync(&p, bytes, st), "hipMallocAsync(packed weights)"); }
    ~AsyncBuf() { if (p) (void)hipFreeAsync(p, st); }
    AsyncBuf(const AsyncBuf&) = delete;
    AsyncBuf& operator=(const AsyncBuf&) = delete;
    dcvc::half_t* half() const { return static_cast<dcvc::half_t*>(p); }
};

struct NsplitPacked {
    int c = 0, ci = 0, device = 0;
    dcvc::half_t* main = nullptr;
    dcvc::half_t* next = nullptr;
    hipEvent_t packed = nullptr;          // recorded behind the pack launches: dcvc_dcb_nsplit_packed on ANOTHER stream waits for it
    hipStream_t pack_stream = nullptr;    // ... the stream they ran on needs no wait (and may be capturing: advisor, round 5)
    mutable bool settled = false;         // the event has been seen complete: no launch waits for it any more
    NsplitPacked() = default;
    NsplitPacked(const NsplitPacked&) = delete;
    NsplitPacked& operator=(const NsplitPacked&) = delete;
    ~NsplitPacked()
    {
        // on the device that owns the buffers, behind every launch that may still read them
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        if (have && cur != device) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        if (main) (void)hipFree(main);
        if (next) (void)hipFree(next);
        if (packed) (void)hipEventDestroy(packed);
        if (have && cur != device) (void)hipSetDevice(cur);
    }
};

void nsplit_check_shape(int c, int ci)
{
    if (!dcvc::dcb_nsplit_shape(c, ci)) {
        throw std::invalid_argument("dcb_nsplit: (block width, inner width) must be (256, 256), (384, 384), (512, 512), (768, 768), (512, 256), (256, 128), (384, 192) or (192, 192)");
    }
}

void nsplit_launch(const dcvc::half_t* wmain, const dcvc::half_t* wnext, const void* t2, int ldt, const void* x, int ldx,
                   const void* b3, const void* b0, const void* b2, const void* q, const void* q2, const void* b1n,
                   void* t1n, int ldt1, void* y, int ldy, int pixels, int c, int ci, int shortcut, hipStream_t st)
{
    dcvc::DcbNsplitDesc d;
    d.t2 = H(t2); d.ldt = ldt; d.x = H(x); d.ldx = ldx;
    d.wmain = wmain;
    d.wnext = wnext;
    d.b3 = H(b3); d.b0 = H(b0); d.b2 = H(b2); d.b1n = H(b1n); d.q = H(q); d.q2 = H(q2);
    d.t1n = H(t1n); d.ldt1 = ldt1; d.y = H(y); d.ldy = ldy;
    d.pixels = pixels; d.c = c; d.ci = ci; d.shortcut = shortcut != 0;
    dcvc::dcb_nsplit(d, st);
}
}  // namespace

int dcvc_dcb_nsplit(const void* t2, int ldt, const void* x, int ldx, const void* w3, const void* b3,
                    const void* w0, const void* b0, const void* w2, const void* b2, const void* q, const void* q2,
                    const void* w1n, const void* b1n, void* t1n, int ldt1, void* y, int ldy,
                    int pixels, int c, int ci, int shortcut, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::kernels_init();
        nsplit_check_shape(c, ci);
        if (!w3 || !w0 || !w2) throw std::invalid_argument("dcb_nsplit: missing operand");
        hipStream_t st = S(stream);
        // packed copies live exactly as long as this call's launches: stream-ordered temporaries, freed on every path out
        const AsyncBuf wmain(dcvc::dcb_nsplit_main_halves(c, ci) * 2, st);
        dcvc::dcb_nsplit_pack_main(H(w3), H(w0), H(w2), c, ci, wmain.half(), st);
        std::unique_ptr<AsyncBuf> wnext;
        if (w1n != nullptr) {
            wnext = std::make_unique<AsyncBuf>(dcvc::dcb_nsplit_dc0_halves(c, ci) * 2, st);
            dcvc::dcb_nsplit_pack_dc0(H(w1n), c, ci, wnext->half(), st);
        }
        nsplit_launch(wmain.half(), wnext ? wnext->half() : nullptr, t2, ldt, x, ldx, b3, b0, b2, q, q2, b1n, t1n, ldt1, y, ldy,
                      pixels, c, ci, shortcut, st);
    });
}

int dcvc_dcb_pair_supported(int cin, int c, int ci)
{
    return dcvc::dcb_pair_supported(cin, c, ci) ? 1 : 0;
}

int dcvc_dcb_pair(const void* x, int ldx, const void* wa, const void* ba, const void* w1, const void* b1,
                  void* y, int ldy, void* t1, int ldt1, int pixels, int cin, int c, int ci, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::kernels_init();
        if (!dcvc::dcb_pair_supported(cin, c, ci)) throw std::invalid_argument("dcb_pair: no kernel variant for this shape");
        if (!x || !wa || !ba || !w1 || !b1 || !y || !t1) throw std::invalid_argument("dcb_pair: missing operand");
        if (x == y) throw std::invalid_argument("dcb_pair: the adaptor output must not alias its input");
        hipStream_t st = S(stream);
        const AsyncBuf pa(dcvc::dcb_pair_adaptor_halves(cin, c) * 2, st);
        dcvc::dcb_pair_pack_adaptor(H(wa), cin, c, pa.half(), st);
        const AsyncBuf p1(dcvc::dcb_nsplit_dc0_halves(c, ci) * 2, st);
        dcvc::dcb_nsplit_pack_dc0(H(w1), c, ci, p1.half(), st);
        dcvc::DcbPairDesc d;
        d.x = H(x); d.ldx = ldx; d.wa = pa.half(); d.ba = H(ba); d.w1 = p1.half(); d.b1 = H(b1);
        d.y = H(y); d.ldy = ldy; d.t1 = H(t1); d.ldt1 = ldt1; d.pixels = pixels; d.cin = cin; d.c = c; d.ci = ci;
        dcvc::dcb_pair(d, st);
    });
}

int dcvc_dcb_nsplit_fin_supported(int c, int ci, int nfin)
{
    return dcvc::dcb_nsplit_fin_supported(c, ci, nfin) ? 1 : 0;
}

int dcvc_dcb_nsplit_fin(const void* t2, int ldt, const void* x, int ldx, const void* w3, const void* b3,
                        const void* w0, const void* b0, const void* w2, const void* b2, const void* q, const void* q2,
                        const void* wfin, const void* bfin, const void* qfin, void* yfin, int ldyfin, int nfin,
                        void* y, int ldy, int pixels, int c, int ci, int shortcut, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::kernels_init();
        nsplit_check_shape(c, ci);
        if (!w3 || !w0 || !w2 || !wfin || !bfin || !yfin) throw std::invalid_argument("dcb_nsplit_fin: missing operand");
        if (!dcvc::dcb_nsplit_fin_supported(c, ci, nfin)) throw std::invalid_argument("dcb_nsplit_fin: no kernel variant for this closing conv");
        hipStream_t st = S(stream);
        const AsyncBuf wmain(dcvc::dcb_nsplit_main_halves(c, ci) * 2, st);
        dcvc::dcb_nsplit_pack_main(H(w3), H(w0), H(w2), c, ci, wmain.half(), st);
        const AsyncBuf wf(dcvc::dcb_nsplit_fin_halves(c, nfin) * 2, st);
        dcvc::dcb_nsplit_pack_fin(H(wfin), c, nfin, wf.half(), st);
        dcvc::DcbNsplitDesc d;
        d.t2 = H(t2); d.ldt = ldt; d.x = H(x); d.ldx = ldx;
        d.wmain = wmain.half();
        d.b3 = H(b3); d.b0 = H(b0); d.b2 = H(b2); d.q = H(q); d.q2 = H(q2);
        d.y = H(y); d.ldy = ldy;
        d.pixels = pixels; d.c = c; d.ci = ci; d.shortcut = shortcut != 0;
        d.wfin = wf.half(); d.bfin = H(bfin); d.qfin = H(qfin); d.yfin = H(yfin); d.ldyfin = ldyfin; d.nfin = nfin;
        dcvc::dcb_nsplit(d, st);
    });
}

int dcvc_dcb_nsplit_dw_supported(int c, int ci, int pixels)
{
    return dcvc::dcb_nsplit_dw_supported(c, ci, pixels) ? 1 : 0;
}

int dcvc_dcb_nsplit_dw(const void* t1, int ldt, const void* wdw, int width, const void* x, int ldx, const void* w3, const void* b3,
                       const void* w0, const void* b0, const void* w2, const void* b2, const void* q, const void* q2,
                       const void* w1n, const void* b1n, void* t1n, int ldt1,
                       const void* wfin, const void* bfin, const void* qfin, void* yfin, int ldyfin, int nfin,
                       void* y, int ldy, int pixels, int c, int ci, int shortcut, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::kernels_init();
        nsplit_check_shape(c, ci);
        if (!t1 || !wdw || !w3 || !w0 || !w2) throw std::invalid_argument("dcb_nsplit_dw: missing operand");
        if (!dcvc::dcb_nsplit_dw_supported(c, ci, pixels)) throw std::invalid_argument("dcb_nsplit_dw: no kernel variant with the depthwise conv inside for this block shape");
        if (wfin != nullptr && !dcvc::dcb_nsplit_fin_supported(c, ci, nfin)) throw std::invalid_argument("dcb_nsplit_dw: no kernel variant for this closing conv");
        if (static_cast<long long>(pixels) * ldt * 2 >= (1LL << 31)) throw std::invalid_argument("dcb_nsplit_dw: dc.0's output must span less than 2 GiB (32-bit byte offsets inside the launch)");
        hipStream_t st = S(stream);
        const AsyncBuf wmain(dcvc::dcb_nsplit_main_halves(c, ci) * 2, st);
        dcvc::dcb_nsplit_pack_main(H(w3), H(w0), H(w2), c, ci, wmain.half(), st);
        std::unique_ptr<AsyncBuf> wnext, wf;
        if (w1n != nullptr) {
            wnext = std::make_unique<AsyncBuf>(dcvc::dcb_nsplit_dc0_halves(c, ci) * 2, st);
            dcvc::dcb_nsplit_pack_dc0(H(w1n), c, ci, wnext->half(), st);
        }
        if (wfin != nullptr) {
            wf = std::make_unique<AsyncBuf>(dcvc::dcb_nsplit_fin_halves(c, nfin) * 2, st);
            dcvc::dcb_nsplit_pack_fin(H(wfin), c, nfin, wf->half(), st);
        }
        dcvc::DcbNsplitDesc d;
        d.t1 = H(t1); d.ldt = ldt; d.wdw = H(wdw); d.width = width; d.x = H(x); d.ldx = ldx;
        d.wmain = wmain.half();
        d.wnext = wnext ? wnext->half() : nullptr;
        d.b3 = H(b3); d.b0 = H(b0); d.b2 = H(b2); d.b1n = H(b1n); d.q = H(q); d.q2 = H(q2);
        d.t1n = H(t1n); d.ldt1 = ldt1; d.y = H(y); d.ldy = ldy;
        d.pixels = pixels; d.c = c; d.ci = ci; d.shortcut = shortcut != 0;
        if (wf) { d.wfin = wf->half(); d.bfin = H(bfin); d.qfin = H(qfin); d.yfin = H(yfin); d.ldyfin = ldyfin; d.nfin = nfin; }
        dcvc::dcb_nsplit(d, st);
    });
}

int dcvc_dcb_nsplit_pack(const void* w3, const void* w0, const void* w2, const void* w1n, int c, int ci, void* stream, void** handle)
{
    return dcvc::guarded([&] {
        dcvc::kernels_init();
        nsplit_check_shape(c, ci);
        if (!w3 || !w0 || !w2 || !handle) throw std::invalid_argument("dcb_nsplit_pack: missing operand");
        auto pk = std::make_unique<NsplitPacked>();          // its destructor frees whatever has been allocated when a step below throws
        pk->c = c; pk->ci = ci;
        dcvc::hip_check(hipGetDevice(&pk->device), "hipGetDevice");
        void* m = nullptr;
        dcvc::hip_check(hipMalloc(&m, dcvc::dcb_nsplit_main_halves(c, ci) * 2), "hipMalloc(packed weights)");
        pk->main = static_cast<dcvc::half_t*>(m);
        dcvc::dcb_nsplit_pack_main(H(w3), H(w0), H(w2), c, ci, pk->main, S(stream));
        if (w1n != nullptr) {
            void* n = nullptr;
            dcvc::hip_check(hipMalloc(&n, dcvc::dcb_nsplit_dc0_halves(c, ci) * 2), "hipMalloc(packed weights)");
            pk->next = static_cast<dcvc::half_t*>(n);
            dcvc::dcb_nsplit_pack_dc0(H(w1n), c, ci, pk->next, S(stream));
        }
        dcvc::hip_check(hipEventCreateWithFlags(&pk->packed, hipEventDisableTiming), "hipEventCreate");
        dcvc::hip_check(hipEventRecord(pk->packed, S(stream)), "hipEventRecord(packed)");
        pk->pack_stream = S(stream);
        *handle = pk.release();
    });
}

int dcvc_dcb_nsplit_free(void* handle)
{
    return dcvc::guarded([&] {
        if (handle == nullptr) return;
        // ~NsplitPacked: synchronises the OWNING device (not whichever is current), then frees
        std::unique_ptr<NsplitPacked> pk(static_cast<NsplitPacked*>(handle));
        // a failure of an earlier launch surfaces at this synchronisation: report it (the destructor cannot), then free anyway
        int cur = 0;
        dcvc::hip_check(hipGetDevice(&cur), "hipGetDevice");
        if (cur != pk->device) dcvc::hip_check(hipSetDevice(pk->device), "hipSetDevice");
        const hipError_t e = hipDeviceSynchronize();
        if (cur != pk->device) (void)hipSetDevice(cur);
        pk.reset();
        dcvc::hip_check(e, "hipDeviceSynchronize(dcb_nsplit_free)");
    });
}

int dcvc_dcb_nsplit_packed(const void* handle, const void* t2, int ldt, const void* x, int ldx, const void* b3, const void* b0,
                           const void* b2, const void* q, const void* q2, const void* b1n, void* t1n, int ldt1,
                           void* y, int ldy, int pixels, int shortcut, int with_next, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::kernels_init();
        if (handle == nullptr) throw std::invalid_argument("dcb_nsplit_packed: null handle");
        const NsplitPacked* pk = static_cast<const NsplitPacked*>(handle);
        int dev = 0;
        dcvc::hip_check(hipGetDevice(&dev), "hipGetDevice");
        if (dev != pk->device) throw std::invalid_argument("dcb_nsplit_packed: the handle was packed on another device");
        if (with_next && pk->next == nullptr) throw std::invalid_argument("dcb_nsplit_packed: packed without the next block's dc.0");
        // the pack launches ran on the stream given to _pack: any OTHER stream orders itself behind them here, until the event has
        // been seen complete once (a wait per launch is a barrier packet per launch; and a stream that is being captured must not
        // wait for an event recorded outside the capture - pack and first launch belong in front of a capture)
        if (S(stream) != pk->pack_stream && !pk->settled) {
            if (hipEventQuery(pk->packed) == hipSuccess) pk->settled = true;
            else dcvc::hip_check(hipStreamWaitEvent(S(stream), pk->packed, 0), "hipStreamWaitEvent(packed)");
        }
        nsplit_launch(pk->main, with_next ? pk->next : nullptr, t2, ldt, x, ldx, b3, b0, b2, q, q2, b1n, t1n, ldt1, y, ldy, pixels,
                      pk->c, pk->ci, shortcut, S(stream));
    });
}

int dcvc_dcb_tail(const void* w1, const void* b1, const void* t, int ldt, const void* dw, const void* x, int ldx,
                  const void* w3, const void* b3,
                  const void* w0, const void* b0, const void* w2, const void* b2, const void* q, const void* q2,
                  void* y, int ldy, int Hh, int W, int c, int cdc, int cffn, int shortcut, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::kernels_init();
        dcvc::DcbTailDesc d;
        d.w1 = H(w1); d.b1 = H(b1); d.t = H(t); d.ldt = ldt; d.dw = H(dw); d.x = H(x); d.ldx = ldx; d.w3 = H(w3); d.b3 = H(b3);
        d.w0 = H(w0); d.b0 = H(b0); d.w2 = H(w2); d.b2 = H(b2); d.q = H(q); d.q2 = H(q2);
        d.y = H(y); d.ldy = ldy; d.H = Hh; d.W = W; d.c = c; d.cdc = cdc; d.cffn = cffn; d.shortcut = shortcut != 0;
        if (d.w1 != nullptr && d.x == d.y) throw std::invalid_argument("dcb_tail with dc.0 inside cannot run in place");
        dcvc::dcb_tail(d, S(stream));
    });
}

int dcvc_scale_clamped(const void* x, int ldx, const void* q, int ldq, void* y, int ldy, int pixels,
                       int C, int reciprocal, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::scale_clamped(H(x), ldx, H(q), ldq, H(y), ldy, pixels, C, reciprocal != 0, S(stream));
    });
}

int dcvc_yuv420_to_x(const void* y, const void* uv, int H_, int W_, void* x, int ldx, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::yuv420_to_x(static_cast<const uint8_t*>(y), static_cast<const uint8_t*>(uv), H_, W_, H(x), ldx,
                          S(stream));
    });
}

int dcvc_x_to_yuv420(const void* x_hat, int row_pixels, int H_, int W_, void* y16, void* uv16, void* y8,
                     void* uv8, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::x_to_yuv420(H(x_hat), row_pixels, H_, W_, H(y16), H(uv16), static_cast<uint8_t*>(y8),
                          static_cast<uint8_t*>(uv8), S(stream));
    });
}

int dcvc_mask_step_enc(void* y, int ldy, const void* q_dec, int ldq, const void* scales, int lds,
                       const void* means, int ldm, void* y_hat, int ldh, void* sym, void* cond,
                       void* block_count, void* compact_out, void* totals, int Hh, int W, int C,
                       int nsteps, int step, float skip_thres, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::symbols_init();
        dcvc::MaskStepEnc d;
        d.y = H(y); d.ldy = ldy; d.q_dec = H(q_dec); d.ldq = ldq; d.scales = H(scales); d.lds = lds;
        d.means = H(means); d.ldm = ldm; d.y_hat = H(y_hat); d.ldh = ldh;
        d.sym = static_cast<int16_t*>(sym); d.cond = static_cast<uint8_t*>(cond);
        d.block_count = static_cast<int32_t*>(block_count);
        d.H = Hh; d.W = W; d.C = C; d.nsteps = nsteps; d.step = step; d.skip_thres = skip_thres;
        dcvc::mask_step_enc(d, S(stream));
        if (step == nsteps - 1) {
            dcvc::compact(sym, 2, d.cond, d.block_count, Hh * W * C, compact_out, static_cast<int32_t*>(totals), 0,
                          S(stream));
        }
    });
}

int dcvc_mask_dec_index(const void* scales, int lds, void* index, void* cond, void* block_count,
                        void* compact_out, void* totals, int Hh, int W, int C, float skip_thres, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::symbols_init();
        dcvc::MaskDecIndex d;
        d.scales = H(scales); d.lds = lds; d.index = static_cast<uint8_t*>(index);
        d.cond = static_cast<uint8_t*>(cond); d.block_count = static_cast<int32_t*>(block_count);
        d.H = Hh; d.W = W; d.C = C; d.skip_thres = skip_thres;
        dcvc::mask_dec_index(d, S(stream));
        dcvc::compact(index, 1, d.cond, d.block_count, Hh * W * C, compact_out, static_cast<int32_t*>(totals), 0,
                      S(stream));
    });
}

int dcvc_mask_step_dec(const void* decoded, const void* cond, const void* block_count, const void* totals,
                       void* yq, const void* means, int ldm, const void* q_dec, int ldq, void* y_hat, int ldh,
                       int Hh, int W, int C, int nsteps, int step, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::MaskStepDec d;
        d.decoded = static_cast<const int8_t*>(decoded); d.cond = static_cast<const uint8_t*>(cond);
        d.block_count = static_cast<const int32_t*>(block_count); d.totals = static_cast<const int32_t*>(totals);
        d.yq = static_cast<int8_t*>(yq); d.means = H(means); d.ldm = ldm; d.q_dec = H(q_dec); d.ldq = ldq;
        d.y_hat = H(y_hat); d.ldh = ldh; d.H = Hh; d.W = W; d.C = C; d.nsteps = nsteps; d.step = step;
        dcvc::mask_step_dec(d, S(stream));
    });
}

int dcvc_dcb_tail_debug_buffer(void* device_buffer)
{
    return dcvc::guarded([&] { dcvc::dcb_tail_debug_buffer(static_cast<dcvc::half_t*>(device_buffer)); });
}

int dcvc_gemm_timeline_buffer(void* device_buffer)
{
    return dcvc::guarded([&] { dcvc::gemm_timeline_buffer(static_cast<long long*>(device_buffer)); });
}

int dcvc_dcb_nsplit_timeline_buffer(void* device_buffer)
{
    return dcvc::guarded([&] { dcvc::dcb_nsplit_timeline_buffer(static_cast<long long*>(device_buffer)); });
}

int dcvc_dcb_nsplit_dw_hook(const void* t1, const void* wdw, int width)
{
    return dcvc::guarded([&] {
        if (t1 != nullptr && (wdw == nullptr || width <= 0)) throw std::invalid_argument("dcb_nsplit_dw_hook: taps and a width");
        dcvc::dcb_nsplit_dw_hook(H(t1), H(wdw), width);
    });
}

int dcvc_round_z(const void* z, void* z_hat, void* z_i8, int count, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::round_z(H(z), H(z_hat), static_cast<int8_t*>(z_i8), count, S(stream));
    });
}

int dcvc_int8_to_half(const void* in, void* out, int count, void* stream)
{
    return dcvc::guarded([&] { dcvc::int8_to_half(static_cast<const int8_t*>(in), H(out), count, S(stream)); });
}

int dcvc_symbol_blocks(int count)
{
    return dcvc::symbol_blocks(count);
}

int dcvc_y_step_enc(const void* y, int ldy, const void* scales, int lds, const void* means, int ldm,
                    void* y_hat_acc, int ldacc, void* sym, void* cond, void* block_count,
                    void* compact_out, void* totals, int Hh, int W, int C, int step,
                    float skip_thres, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::symbols_init();
        dcvc::YStepEnc d;
        d.y = H(y); d.ldy = ldy; d.scales = H(scales); d.lds = lds; d.means = H(means); d.ldm = ldm;
        d.y_hat_acc = H(y_hat_acc); d.ldacc = ldacc;
        d.sym = static_cast<int16_t*>(sym); d.cond = static_cast<uint8_t*>(cond);
        d.block_count = static_cast<int32_t*>(block_count);
        d.H = Hh; d.W = W; d.C = C; d.step = step; d.skip_thres = skip_thres; d.first = (step == 0);
        dcvc::y_step_enc(d, S(stream));
        dcvc::compact(sym, 2, d.cond, d.block_count, Hh * W * (C / 4), compact_out,
                      static_cast<int32_t*>(totals), step, S(stream));
    });
}

int dcvc_y_step_dec_index(const void* scales, int lds, void* index, void* cond, void* block_count,
                          void* compact_out, void* totals, int Hh, int W, int C, int step,
                          float skip_thres, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::symbols_init();
        dcvc::YStepDecIndex d;
        d.scales = H(scales); d.lds = lds;
        d.index = static_cast<uint8_t*>(index); d.cond = static_cast<uint8_t*>(cond);
        d.block_count = static_cast<int32_t*>(block_count);
        d.H = Hh; d.W = W; d.C = C; d.step = step; d.skip_thres = skip_thres;
        dcvc::y_step_dec_index(d, S(stream));
        dcvc::compact(index, 1, d.cond, d.block_count, Hh * W * (C / 4), compact_out,
                      static_cast<int32_t*>(totals), step, S(stream));
    });
}

int dcvc_y_step_dec_restore(const void* decoded, const void* cond, const void* block_count,
                            const void* totals, const void* means, int ldm, void* y_hat_acc,
                            int ldacc, int Hh, int W, int C, int step, void* stream)
{
    return dcvc::guarded([&] {
        dcvc::YStepDecRestore d;
        d.decoded = static_cast<const int8_t*>(decoded);
        d.cond = static_cast<const uint8_t*>(cond);
        d.block_count = static_cast<const int32_t*>(block_count);
        d.totals = static_cast<const int32_t*>(totals); d.slot = step;
        d.means = H(means); d.ldm = ldm; d.y_hat_acc = H(y_hat_acc); d.ldacc = ldacc;
        d.H = Hh; d.W = W; d.C = C; d.step = step; d.first = (step == 0);
        dcvc::y_step_dec_restore(d, S(stream));
    });
}

int dcvc_gemm_profile_enable(int on)
{
    return dcvc::guarded([&] { dcvc::gemm_profile_enable(on != 0); });
}

int dcvc_gemm_profile_reset(void)
{
    return dcvc::guarded([&] { dcvc::gemm_profile_reset(); });
}

int dcvc_gemm_profile_collect(double* ms, double* flops, long long* launches)
{
    return dcvc::guarded([&] { dcvc::gemm_profile_collect(ms, flops, launches); });
}

long long dcvc_gemm_profile_launches(void* records, long long cap)
{
    long long n = -1;
    dcvc::guarded([&] {
        n = static_cast<long long>(dcvc::gemm_profile_launches(static_cast<dcvc::GemmLaunchInfo*>(records),
                                                                 static_cast<size_t>(cap)));
    });
    return n;
}


}  // extern "C"
