// Host side of dcb_pair8_kernel.h: the adaptor's weight stream, shape dispatch.
#include "dcb_pair8_kernel.h"

namespace dcvc {

int dcb_nsplit_waves();

namespace {

// [C][CIN] -> per-wave streams of MFMA "A" fragments in the order the kernel consumes them (k-slice major, the wave's tiles
// inside): waves 0 .. 3 (HI_C tiles each: tiles [simd QC, simd QC + HI_C)), then 4 .. 7 (the remaining LO_C of the SIMD pair)
__global__ void pack_adaptor8_kernel(const half_t* w, int C, int CIN, half8* out)
{
    // tiles per wave as Geo<>: by SIMD pair for C a multiple of 128, else (192) waves 0 .. 3 one tile each and the first `act` of
    // the waves 4 .. 7 one each - the others get a tile of zeros
    const int KS = CIN / 16;
    const bool by_pair = C % 128 == 0;
    int hi, lo, act;
    if (by_pair) { const int q = C / 128; hi = (q + 1) / 2; lo = q / 2; act = 4; }
    else { const int tn = C / 32; hi = (tn + 7) / 8; const int rem = tn - 4 * hi; lo = rem <= 0 ? 0 : (rem + 3) / 4; act = lo == 0 ? 0 : rem / lo; }
    const int FA_HI = hi * KS, FA_LO = lo * KS;
    const long long u = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (u >= 4LL * (FA_HI + FA_LO) * 64) return;
    const int lane = static_cast<int>(u & 63);
    const int F = static_cast<int>(u >> 6);
    const bool hiw = F < 4 * FA_HI;
    const int wave = hiw ? F / FA_HI : 4 + (F - 4 * FA_HI) / FA_LO;
    const int f = hiw ? F % FA_HI : (F - 4 * FA_HI) % FA_LO;
    const int NT = hiw ? hi : lo;
    const int first = by_pair ? (wave & 3) * (C / 128) + (hiw ? 0 : hi) : (hiw ? wave * hi : 4 * hi + (wave - 4) * lo);
    const int ks = f / NT, n0 = 32 * (first + f % NT);
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (by_pair || hiw || wave - 4 < act) v = *reinterpret_cast<const half8*>(w + static_cast<size_t>(n0 + (lane & 31)) * CIN + 16 * ks + 8 * (lane >> 5));
    out[u] = v;
}

}  // namespace

namespace pair8 {
// dcb_pair8_<C>.hip
void run_c192(const PairParams& p, int cin, int ci, bool wide, hipStream_t stream);
void run_c256(const PairParams& p, int cin, int ci, bool wide, hipStream_t stream);
void run_c384(const PairParams& p, int cin, int ci, bool wide, hipStream_t stream);
void run_c512(const PairParams& p, int cin, int ci, bool wide, hipStream_t stream);
}  // namespace pair8

bool dcb_pair_supported(int cin, int c, int ci)
{
    static const bool off = [] { const char* e = getenv("DCVC_PAIR"); return e != nullptr && atoi(e) == 0; }();    // A/B: adaptor and dc.0 as two launches
    if (off || dcb_nsplit_waves() != 8) return false;
    if (c == 256 && ci == 128) return cin == 448 || cin == 512 || cin == 192;       // LD: encoder, decoder / adaptor_m / spatial prior, adaptor_i
    if (c == 256 && ci == 256) return cin == 128 || cin == 512;                     // intra hyper decoder; hierarchical reconstruction heads
    if (c == 384 && ci == 384) return cin == 192;                                   // intra encoder
    if (c == 192 && ci == 192) return cin == 384;                                   // intra decoder's last block
    if (c == 512 && ci == 512) return cin == 256 || cin == 512 || cin == 192;       // prior fusion, spatial prior adaptors, HT-L adaptor_i
    if (c == 512 && ci == 256) return cin == 192;                                   // HT-S adaptor_i
    return false;
}

size_t dcb_pair_adaptor_halves(int cin, int c)
{
    // (block width 192: + the zero tiles of the two waves without a share)
    const int tiles = c % 128 == 0 ? c / 32 : 8 * ((c / 32 + 7) / 8);
    return 1ull * tiles * (cin / 16) * 512;
}

void dcb_pair_pack_adaptor(const half_t* wa, int cin, int c, half_t* out, hipStream_t stream)
{
    if (c % 64 != 0 || cin % 64 != 0) throw std::invalid_argument("dcb_pair: unsupported adaptor shape");
    const long long units = static_cast<long long>(dcb_pair_adaptor_halves(cin, c) / 8);
    hipLaunchKernelGGL(pack_adaptor8_kernel, dim3(static_cast<unsigned>((units + 255) / 256)), dim3(256), 0, stream, wa, c, cin,
                       reinterpret_cast<half8*>(out));
    hip_check(hipGetLastError(), "dcb_pair pack");
}

void dcb_pair(const DcbPairDesc& d, hipStream_t stream)
{
    if (!dcb_pair_supported(d.cin, d.c, d.ci)) throw std::invalid_argument("dcb_pair: no kernel for this (input, block, inner) width");
    if (d.pixels <= 0) throw std::invalid_argument("dcb_pair: empty problem");
    if ((d.ldx % 8) || (d.ldy % 8) || (d.ldt1 % 8)) throw std::invalid_argument("dcb_pair: leading dimensions must be multiples of 8 channels");
    if (!d.x || !d.wa || !d.ba || !d.w1 || !d.b1 || !d.y || !d.t1) throw std::invalid_argument("dcb_pair: missing operand");
    pair8::PairParams p{};
    p.x = d.x; p.ldx = d.ldx; p.wa = reinterpret_cast<const half8*>(d.wa); p.w1 = reinterpret_cast<const half8*>(d.w1);
    p.ba = d.ba; p.b1 = d.b1; p.wsilu = wsilu_table_device();
    p.y = d.y; p.ldy = d.ldy; p.t1 = d.t1; p.ldt1 = d.ldt1; p.M = d.pixels;
    const bool wide = d.pixels >= 64 * 200;       // as the block kernel: 64-pixel workgroups when they fill the chip
    if (d.c == 192) pair8::run_c192(p, d.cin, d.ci, wide, stream);
    else if (d.c == 256) pair8::run_c256(p, d.cin, d.ci, wide, stream);
    else if (d.c == 384) pair8::run_c384(p, d.cin, d.ci, wide, stream);
    else pair8::run_c512(p, d.cin, d.ci, wide, stream);
}

}  // namespace dcvc
