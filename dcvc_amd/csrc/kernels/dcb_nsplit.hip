// Host side of the N-split DepthConvBlock kernel (dcb_nsplit8_kernel.h): weight packing, shape dispatch.
#include "dcb_nsplit8_kernel.h"

namespace dcvc {

using namespace nsplit;

namespace {

// ------------------------------------------------------------------------------------ weight packing
// [waves][fragments][64 lanes][8 halves]: fragment = the MFMA "A" operand of one (32-channel tile, 16-deep k-slice):
// lane l holds row (l & 31), k = 8 (l >> 5) .. + 7. Order inside a wave's stream = the order the kernel consumes.
// the kernel's streams (dcb_nsplit8_kernel.h): waves 0 .. 3, then 4 .. 7; waves w and w + 4 share the tiles
// [simd QC, (simd + 1) QC) of a C-wide layer (w the first HI, w + 4 the remaining LO); ffn.0: N0 tiles per wave in passes of 2
// Geo<> on the host: tiles per wave of an N-wide layer. N a multiple of 128: by SIMD pair (HI = first share, LO = the rest);
// otherwise (192) by the closing convs' rule: waves 0 .. 3 `hi` tiles each, the first `act` of the waves 4 .. 7 `lo` each.
struct Share {
    int hi, lo, act;
    bool by_pair;
    __host__ __device__ explicit Share(int n)
    {
        by_pair = n % 128 == 0;
        if (by_pair) {
            const int q = n / 128;
            hi = (q + 1) / 2; lo = q / 2; act = 4;
        } else {
            const int tn = n / 32;
            hi = (tn + 7) / 8;
            const int rem = tn - 4 * hi;
            lo = rem <= 0 ? 0 : (rem + 3) / 4;
            act = lo == 0 ? 0 : rem / lo;
        }
    }
    // first tile of wave `w` (0 .. 7) and whether it has a share at all
    __host__ __device__ int first_tile(int w, int n) const
    {
        if (by_pair) return (w & 3) * (n / 128) + (w < 4 ? 0 : hi);
        return w < 4 ? w * hi : 4 * hi + (w - 4) * lo;
    }
    __host__ __device__ bool on(int w) const { return by_pair || w < 4 || w - 4 < act; }
};

__global__ void pack_main8_kernel(const half_t* w3, const half_t* w0, const half_t* w2, int C, int CI, half8* out)
{
    const int KS_C = C / 16, KS_I = CI / 16, TP = 2;
    const Share sc(C);
    const int PAIRS = CI / 16, P0_HI = (PAIRS + 7) / 8, P0_LO = (PAIRS - 4 * P0_HI) / 4;       // Geo<>: ffn.0 tile pairs per wave (CI = 192: 2 | 1)
    const int FM_HI = 2 * sc.hi * KS_I + 2 * P0_HI * KS_C, FM_LO = 2 * sc.lo * KS_I + 2 * P0_LO * KS_C;
    const long long u = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (u >= 4LL * (FM_HI + FM_LO) * 64) return;
    const int lane = static_cast<int>(u & 63);
    const int F = static_cast<int>(u >> 6);
    const bool hiw = F < 4 * FM_HI;
    const int wave = hiw ? F / FM_HI : 4 + (F - 4 * FM_HI) / FM_LO;
    const int f = hiw ? F % FM_HI : (F - 4 * FM_HI) % FM_LO;
    const int NT = hiw ? sc.hi : sc.lo, F_DC3 = NT * KS_I, F_FFN0 = 2 * (hiw ? P0_HI : P0_LO) * KS_C;
    const int cb = 32 * sc.first_tile(wave, C);
    const int cb0 = hiw ? wave * 64 * P0_HI : 4 * 64 * P0_HI + (wave - 4) * 64 * P0_LO;       // the wave's first ffn.0 channel
    const half_t* w;
    int n0, ks, K;
    bool zero = false;             // a wave without a share of the C-wide layers walks a tile of zeros (block width 192: waves 6, 7)
    if (f < F_DC3) {                                  // dc.3 [C][CI]
        ks = f / NT; n0 = cb + 32 * (f % NT); w = w3; K = CI; zero = !sc.on(wave);
    } else if (f < F_DC3 + F_FFN0) {                  // ffn.0 [4 CI][C]
        const int g = f - F_DC3, pass = g / (TP * KS_C), r = g % (TP * KS_C);
        ks = r / TP; n0 = cb0 + pass * 32 * TP + 32 * (r % TP); w = w0; K = C;
    } else {                                          // ffn.2 [C][CI]
        const int g = f - F_DC3 - F_FFN0;
        ks = g / NT; n0 = cb + 32 * (g % NT); w = w2; K = CI; zero = !sc.on(wave);
    }
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!zero) v = *reinterpret_cast<const half8*>(w + static_cast<size_t>(n0 + (lane & 31)) * K + 16 * ks + 8 * (lane >> 5));
    out[u] = v;
}

__global__ void pack_dc08_kernel(const half_t* w1, int C, int CI, half8* out)       // dc.0 [CI][C]
{
    const int KS_C = C / 16, QI = CI / 128, HI_I = (QI + 1) / 2, LO_I = QI / 2, FD_HI = HI_I * KS_C, FD_LO = LO_I * KS_C;
    const long long u = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (u >= 4LL * (FD_HI + FD_LO) * 64) return;
    const int lane = static_cast<int>(u & 63);
    const int F = static_cast<int>(u >> 6);
    const bool hiw = F < 4 * FD_HI;
    const int wave = hiw ? F / FD_HI : 4 + (F - 4 * FD_HI) / FD_LO;
    const int f = hiw ? F % FD_HI : (F - 4 * FD_HI) % FD_LO;
    const int simd = wave & 3, NT = hiw ? HI_I : LO_I;
    const int ks = f / NT, n0 = 32 * (simd * QI + (hiw ? 0 : HI_I) + f % NT);
    out[u] = *reinterpret_cast<const half8*>(w1 + static_cast<size_t>(n0 + (lane & 31)) * C + 16 * ks + 8 * (lane >> 5));
}

// the closing conv of a chain in the 8-wave kernel's NEXT slot, [NN][C]: waves 0 .. 3 own nf_hi tiles each, of the waves 4 .. 7
// the first act_lo own nf_lo each (Geo<>::nf_hi / nf_lo / act_lo); the streams of idle waves stay zero
__global__ void pack_fin8_kernel(const half_t* w, int C, int NN, half8* out)
{
    const int KS_C = C / 16, TN = NN / 32, HI = (TN + 7) / 8, REM = TN - 4 * HI, LO = REM <= 0 ? 0 : (REM + 3) / 4, ACT = LO == 0 ? 0 : REM / LO;
    const int FD_HI = HI * KS_C, FD_LO = LO * KS_C;
    const long long u = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (u >= 4LL * (FD_HI + FD_LO) * 64) return;
    const int lane = static_cast<int>(u & 63);
    const int F = static_cast<int>(u >> 6);
    const bool hiw = F < 4 * FD_HI;
    const int wave = hiw ? F / FD_HI : 4 + (F - 4 * FD_HI) / FD_LO;
    const int f = hiw ? F % FD_HI : (F - 4 * FD_HI) % FD_LO;
    const int NT = hiw ? HI : LO;
    const int ks = f / NT, tile = (hiw ? wave * HI : 4 * HI + (wave - 4) * LO) + f % NT;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hiw || wave - 4 < ACT) v = *reinterpret_cast<const half8*>(w + static_cast<size_t>(32 * tile + (lane & 31)) * C + 16 * ks + 8 * (lane >> 5));
    out[u] = v;
}

long long* g_ns_timeline = nullptr;
// tuning aid (tools/probes/core_bench.hip -w): launches of a shape that has the variant run it with the depthwise conv inside, on these operands
const half_t* g_dw_hook_t1 = nullptr;
const half_t* g_dw_hook_taps = nullptr;
int g_dw_hook_width = 0;

}  // namespace

bool dcb_nsplit_shape(int c, int ci)
{
    return (c == 384 && ci == 384) || (c == 512 && ci == 512) || (c == 768 && ci == 768) || (c == 256 && ci == 256) ||
           (c == 512 && ci == 256) || (c == 256 && ci == 128) ||
           (c == 384 && ci == 192) ||     // round 6: the LD model's prior fusion blocks
           (c == 192 && ci == 192);       // ... and the intra decoder's last block
}

// widths of a chain-closing conv the 8-wave kernel of a block shape is instantiated for (dcb_nsplit8_<shape>_fin.hip)
bool dcb_nsplit_fin_supported(int c, int ci, int nn)
{
    static const bool off = [] { const char* e = getenv("DCVC_NSPLIT_FIN"); return e != nullptr && atoi(e) == 0; }();   // A/B: the closing convs as launches of their own
    if (off || !dcb_nsplit_shape(c, ci)) return false;
    if (c == 256 && ci == 128) return nn == 128 || nn == 192 || nn == 256;
    if (c == 256 && ci == 256) return nn == 192;
    if (c == 512 && ci == 512) return nn == 256 || nn == 512;
    if (c == 768 && ci == 768) return nn == 768;
    if (c == 384 && ci == 192) return nn == 384;
    return false;
}

// block shapes the kernel is also instantiated for with the block's depthwise conv inside (dcb_nsplit8_256_128_dw.hip: LDS has room
// for dc.0's output around a tile next to the activations only in the narrow blocks)
static bool nsplit_wide(int pixels, int c)
{
    // 64-pixel workgroups when they fill the chip (picture resolution / 8), 32 otherwise (/ 16: 255 workgroups at 1080p)
    // (768-wide blocks - the hierarchical models' prior fusion at / 16 - have LDS for 32 pixels only)
    // DCVC_NSPLIT_PX=32: 32-pixel workgroups everywhere (A/B: twice the tiles per workgroup, half the work per weight byte)
    static const bool narrow_all = [] { const char* e = getenv("DCVC_NSPLIT_PX"); return e != nullptr && atoi(e) == 32; }();
    return pixels >= 64 * 200 && c < 768 && !narrow_all;
}

bool dcb_nsplit_dw_supported(int c, int ci, int pixels)
{
    static const bool off = [] { const char* e = getenv("DCVC_NSPLIT_DW"); return e != nullptr && atoi(e) == 0; }();   // A/B: the depthwise conv as a launch of its own
    if (off || pixels <= 0) return false;
    return (c == 256 && ci == 128) || (c == 384 && ci == 192 && !nsplit_wide(pixels, c));
}

void dcb_nsplit_dw_hook(const half_t* t1, const half_t* wdw, int width)
{
    g_dw_hook_t1 = t1; g_dw_hook_taps = wdw; g_dw_hook_width = width;
}

void dcb_nsplit_timeline_buffer(long long* device_buffer)
{
    g_ns_timeline = device_buffer;
}

// fragments of 512 halves: dc.3 and ffn.2 (the waves' tiles x ci / 16 slices each - incl. the zero tiles of waves without a share),
// ffn.0 (4 ci / 32 tiles x c / 16 slices)
size_t dcb_nsplit_main_halves(int c, int ci)
{
    const Share sc(c);
    return (2ull * 4 * (sc.hi + sc.lo) * (ci / 16) + 1ull * (ci / 8) * (c / 16)) * 512;
}
size_t dcb_nsplit_dc0_halves(int c, int ci)
{
    // an inner width that is no multiple of 128 (192): tiles by wave as a closing conv's, with the streams of the idle waves
    return ci % 128 == 0 ? 1ull * (ci / 32) * (c / 16) * 512 : dcb_nsplit_fin_halves(c, ci);
}

size_t dcb_nsplit_fin_halves(int c, int nn)      // (declared in ops.h)
{
    const int tn = nn / 32, hi = (tn + 7) / 8, rem = tn - 4 * hi, lo = rem <= 0 ? 0 : (rem + 3) / 4;
    return 4ull * (hi + lo) * (c / 16) * 512;
}

void dcb_nsplit_pack_fin(const half_t* w, int c, int nn, half_t* out, hipStream_t stream)
{
    if (nn % 32 != 0 || nn < 128 || c % 64 != 0) throw std::invalid_argument("dcb_nsplit: unsupported closing conv");
    const long long units = static_cast<long long>(dcb_nsplit_fin_halves(c, nn) / 8);
    hipLaunchKernelGGL(pack_fin8_kernel, dim3(static_cast<unsigned>((units + 255) / 256)), dim3(256), 0, stream, w, c, nn, reinterpret_cast<half8*>(out));
    hip_check(hipGetLastError(), "dcb_nsplit pack");
}

void dcb_nsplit_pack_main(const half_t* w3, const half_t* w0, const half_t* w2, int c, int ci, half_t* out, hipStream_t stream)
{
    if (!dcb_nsplit_shape(c, ci)) throw std::invalid_argument("dcb_nsplit: unsupported block shape");
    const long long units = static_cast<long long>(dcb_nsplit_main_halves(c, ci) / 8);
    hipLaunchKernelGGL(pack_main8_kernel, dim3(static_cast<unsigned>((units + 255) / 256)), dim3(256), 0,
                       stream, w3, w0, w2, c, ci, reinterpret_cast<half8*>(out));
    hip_check(hipGetLastError(), "dcb_nsplit pack");
}

void dcb_nsplit_pack_dc0(const half_t* w1, int c, int ci, half_t* out, hipStream_t stream)
{
    if (!dcb_nsplit_shape(c, ci)) throw std::invalid_argument("dcb_nsplit: unsupported block shape");
    const long long units = static_cast<long long>(dcb_nsplit_dc0_halves(c, ci) / 8);
    if (ci % 128 != 0) {
        dcb_nsplit_pack_fin(w1, c, ci, out, stream);
        return;
    }
    hipLaunchKernelGGL(pack_dc08_kernel, dim3(static_cast<unsigned>((units + 255) / 256)), dim3(256), 0,
                       stream, w1, c, ci, reinterpret_cast<half8*>(out));
    hip_check(hipGetLastError(), "dcb_nsplit pack");
}

int dcb_nsplit_waves()
{
    return 8;       // (round 3's 4-wave form, DCVC_NSPLIT_WAVES=4, was the A/B partner through rounds 4 and 5: retired in round 6)
}

int dcb_nsplit_mode()
{
    // DCVC_NSPLIT: 0 = never (A/B against dcb_core / dcb_tail / the launch sequence), 1 = full-width blocks only,
    // unset / 2 = wherever the shape allows (the half-width `dcb2` blocks of the inter models too)
    static const int mode = [] { const char* e = getenv("DCVC_NSPLIT"); return e != nullptr ? atoi(e) : 2; }();
    return mode;
}

bool dcb_nsplit_supported(int c, int cdc, int cffn)
{
    static const bool core_off = [] { const char* e = getenv("DCVC_NO_DCB_CORE"); return e != nullptr && atoi(e) != 0; }();
    if (core_off || dcb_nsplit_mode() == 0 || cdc != cffn || !dcb_nsplit_shape(c, cdc)) return false;
    return cdc == c || dcb_nsplit_mode() >= 2;
}

void dcb_nsplit(const DcbNsplitDesc& desc, hipStream_t stream)
{
    DcbNsplitDesc d = desc;
    if (g_dw_hook_t1 != nullptr && d.t1 == nullptr && dcb_nsplit_dw_supported(d.c, d.ci, d.pixels) && d.pixels % g_dw_hook_width == 0) {
        d.t2 = nullptr; d.t1 = g_dw_hook_t1; d.wdw = g_dw_hook_taps; d.width = g_dw_hook_width;
    }
    if (!dcb_nsplit_shape(d.c, d.ci)) {
        throw std::invalid_argument("dcb_nsplit: (block width, inner width) must be (256, 256), (384, 384), (512, 512), (768, 768), (512, 256), (256, 128), (384, 192) or (192, 192)");
    }
    if (d.pixels <= 0) throw std::invalid_argument("dcb_nsplit: empty problem");
    if ((d.ldt % 8) || (d.ldx % 8) || (d.ldy % 8) || (d.wnext && d.ldt1 % 8)) {
        throw std::invalid_argument("dcb_nsplit: leading dimensions must be multiples of 8 channels");
    }
    if (d.t1 != nullptr) {
        if (d.t2 != nullptr) throw std::invalid_argument("dcb_nsplit: either the depthwise conv's output (t2) or its input (t1)");
        if (!dcb_nsplit_dw_supported(d.c, d.ci, d.pixels)) throw std::invalid_argument("dcb_nsplit: no kernel with the depthwise conv inside for this block shape and size");
        if (!d.wdw || d.width <= 0 || d.pixels % d.width != 0) throw std::invalid_argument("dcb_nsplit: depthwise conv inside needs its taps and the picture's width");
        if (d.t1 == d.t1n) throw std::invalid_argument("dcb_nsplit: dc.0's output for the next block must not overwrite this block's (neighbouring workgroups read it)");
    }
    if ((!d.t2 && !d.t1) || !d.x || !d.wmain || !d.b3 || !d.b0 || !d.b2 || (!d.y && !d.wfin) || (d.wnext && (!d.b1n || !d.t1n))) {
        throw std::invalid_argument("dcb_nsplit: missing operand");
    }
    if (d.wfin != nullptr) {
        if (d.wnext != nullptr) throw std::invalid_argument("dcb_nsplit: a block either hands dc.0 to the next block or closes the chain");
        if (!dcb_nsplit_fin_supported(d.c, d.ci, d.nfin)) throw std::invalid_argument("dcb_nsplit: no kernel for this closing conv");
        if (!d.bfin || !d.yfin || d.ldyfin % 8) throw std::invalid_argument("dcb_nsplit: closing conv needs bias, output and a leading dimension in units of 8");
    }
    NsParams p{};
    p.t2 = d.t2; p.ldt = d.ldt; p.x = d.x; p.ldx = d.ldx;
    p.t1 = d.t1; p.wdw = d.wdw; p.W = d.width;
    p.wmain = reinterpret_cast<const half8*>(d.wmain); p.wnext = reinterpret_cast<const half8*>(d.wnext);
    p.b3 = d.b3; p.b0 = d.b0; p.b2 = d.b2; p.b1n = d.b1n; p.q = d.q; p.q2 = d.q2;
    p.wsilu = wsilu_table_device();
    p.y = d.y; p.ldy = d.ldy; p.t1n = d.t1n; p.ldt1 = d.ldt1; p.M = d.pixels; p.shortcut = d.shortcut ? 1 : 0;
    p.timeline = g_ns_timeline;
    if (d.wfin != nullptr) {      // the NEXT slot holds the chain's closing conv
        p.wnext = reinterpret_cast<const half8*>(d.wfin); p.b1n = d.bfin; p.qf = d.qfin; p.t1n = d.yfin; p.ldt1 = d.ldyfin;
    }
    const bool wide = nsplit_wide(d.pixels, d.c);
    const int next = d.wfin != nullptr ? d.nfin : d.wnext != nullptr ? 1 : 0;
    if (d.t1 != nullptr && d.c == 384) nsplit8::run_384_192_dw(p, next, stream);
    else if (d.t1 != nullptr) nsplit8::run_256_128_dw(p, wide, next, stream);
    else if (d.c == 192) nsplit8::run_192_192(p, wide, next, stream);
    else if (d.c == 384 && d.ci == 192) nsplit8::run_384_192(p, wide, next, stream);
    else if (d.c == 384) nsplit8::run_384_384(p, wide, next, stream);
    else if (d.c == 768) nsplit8::run_768_768(p, wide, next, stream);
    else if (d.c == 512 && d.ci == 512) nsplit8::run_512_512(p, wide, next, stream);
    else if (d.c == 512) nsplit8::run_512_256(p, wide, next, stream);
    else if (d.ci == 256) nsplit8::run_256_256(p, wide, next, stream);
    else nsplit8::run_256_128(p, wide, next, stream);
}

}  // namespace dcvc
