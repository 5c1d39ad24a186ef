// dcb_nsplit8_kernel.h - the N-split DepthConvBlock kernel: a DepthConvBlock behind its depthwise conv in ONE launch,
// activations in LDS, EIGHT waves per workgroup - two per SIMD, 256 registers each, every wave owning an EIGHTH of a layer's
// output channels and streaming ITS weights from L2 (round 3 introduced the N-split form with four waves; round 4 this one).
//
//     y1 = W3 * t2 + b3' + x                          dc.3 (+ folded depthwise bias) + block input
//     t  = chunk_add(WSiLU(W0 * y1 + b0))             ffn.0   (4x expansion, never materialised)
//     y  = (W2 * t + b2 + y1 [+ x]) [* q] -> fp16 [* q2]     ffn.2 (+ block shortcut, + quant scales)
//     [t1' = WSiLU(W1' * y + b1')]                    dc.0 of the NEXT block of a chain (optional, NEXT = 1)
//     [o   = (Wf * y + bf) [* qf] -> fp16]            or the 1x1 conv that CLOSES the chain (optional, NEXT = its width NN:
//                                                     y_prior_fusion.conv.3, y_spatial_prior.conv.3, decoder.conv2 with its
//                                                     quant scale, recon_head.head ...; round 6)
//
// Reference: DepthConvBlockProxy::forward, layers_proxy.cpp:71-101; the chain-closing convs dmci_proxy.cpp:145-199,
// dmc_ld_proxy.cpp:420-593 (conv1x1_bias / conv1x1_bias_with_quant launches of their own there). Same contract, same arithmetic (contraction order,
// bias-initialised accumulators, epilogue order, rounding points) as dcb_nsplit / conv_gemm: bit-identical
// (tests/test_kernels_gpu.py) and equal to the oracle.
//
// Why (round 4, profiles/r04_phase_overlap.txt, tools/probes/phase_overlap.hip): a lone in-order wave per SIMD - round
// 3's kernel - runs its MFMAs and its epilogue arithmetic one after the other: cutting an epilogue into pieces between
// its own MFMAs hid 13 - 40 % of it, and a third of the tile was epilogue time with the matrix core idle. TWO waves on
// a SIMD that each alternate [contraction] [own epilogue] fall out of step by themselves (they compete for the one
// matrix pipe) and then overlap: the probe's 96 MFMAs + 600 VALU per SIMD take 5 914 cycles on one wave and 3 793 on
// two; with LDS gathers in the epilogue 23.0 k against 11.9 k. So here:
//   * 8 waves; waves w and w + 4 share a SIMD (a workgroup's waves are dealt out cyclically over the four SIMDs) and
//     between them own a QUARTER of a layer's 32-channel tiles - split evenly where the quarter is even, 2 + 1 for the
//     384-wide layers (wave w the larger share), 1 + 0 for dc.0 of the (256, 128) blocks; ffn.0's 4 CI channels split
//     evenly over all eight waves, in passes of 2 tiles;
//   * every wave streams ITS weight fragments (packed per wave, dcb_nsplit.hip) into an 8-fragment register ring by
//     buffer_load; the workgroup as a whole still reads each weight byte once per 64 pixels;
//   * no software pipelining inside a wave: a pass is [MFMAs] [its WSiLU + chunk-add epilogue], the partner wave fills
//     the matrix pipe meanwhile. One accumulator set: 256 registers suffice.
// Otherwise: activations in LDS (A = [PX][CI], B = [PX][C], XOR-swizzled), persistent workgroups,
// the next tile's t2 by LDS-DMA behind ffn.2's MFMAs, outputs stored straight from the epilogues' registers.
#pragma once
#include "dcb_nsplit_common.h"

namespace dcvc {
namespace nsplit8 {

using nsplit::align16k;
using nsplit::lds_dma16;
using nsplit::NsParams;
using nsplit::R;
using nsplit::static_for;
using nsplit::TABLE_BYTES;
using nsplit::uint4v;

constexpr int NTHREADS = 512;
#ifndef NS8_RING
#define NS8_RING 8
#endif
constexpr int RING = NS8_RING;           // weight fragments in flight per wave (4 registers each): 64-pixel workgroups
// 32-pixel workgroups (PXT = 1: the prior networks at picture resolution / 16) have the registers for a deeper ring, and need it:
// a fragment feeds ONE MFMA there, the weight stream per workgroup is the bound (round 6: profiles/r06_ring_px32.txt)
#ifndef NS8_RING1
#define NS8_RING1 8
#endif
// how many k-slices ahead of their MFMAs the activation fragments are read from LDS: NS8_AHEAD where a slice has fewer than
// four MFMAs (one tile per wave, or 32-pixel workgroups), NS8_AHEAD_WIDE where it has four or more
#ifndef NS8_AHEAD
#define NS8_AHEAD 2
#endif
#ifndef NS8_AHEAD_WIDE
#define NS8_AHEAD_WIDE 1
#endif
#ifndef NS8_GATHERS
#define NS8_GATHERS 8
#endif
constexpr int GB = NS8_GATHERS;          // table gathers in flight in the WSiLU epilogues (4 registers each)

// LDS layout. TRIPLE (NS8_TRIPLE, where it fits): a SECOND t2 buffer A' behind B, so that the next tile's t2 can land while the
// current tile still works in A - its transfer goes out in front of the last ffn.0 pass's epilogue (no weights are requested
// behind it for a while) instead of behind a barrier of its own after ffn.2's MFMAs. A and A' swap roles every tile. The 384-wide
// blocks have room for it only with ONE copy of the WSiLU table (3 x 48 KB + 4 KB + constants = exactly 160 KB).
#ifndef NS8_TRIPLE
#define NS8_TRIPLE 1
#endif
template <int C, int CI, int PXT, int NEXT = 1, int DW = 0>
struct Lay {
    // LDS rows are swizzled in groups of 16 chunks (256 bytes): an inner width of 192 (the LD model's (384, 192) prior
    // fusion blocks) lives in rows of 256 channels, a third of them never read
    static constexpr int CIP = (CI + 127) / 128 * 128;
    static constexpr int CP = (C + 127) / 128 * 128;                     // (likewise the C-wide tensors: the intra decoder's last block is 192 wide)
    static constexpr int BUF_A = 32 * PXT * CIP * 2, BUF_B = 32 * PXT * CP * 2;
    static constexpr int NB = NEXT > 1 ? NEXT : CI;                      // bias floats of the NEXT slot (dc.0: CI, closing conv: NN)
    static constexpr int BIAS_FLOATS = 2 * C + 4 * CI + NB;
    static constexpr int CONSTS = BIAS_FLOATS * 4 + 2 * C * 2;
    static constexpr bool FIT4 = 2 * BUF_A + BUF_B + 4 * TABLE_BYTES + CONSTS <= 160 * 1024 && (2 * BUF_A + BUF_B) % 16384 == 0;
    static constexpr bool FIT1 = 2 * BUF_A + BUF_B + 1 * TABLE_BYTES + CONSTS <= 160 * 1024 && (2 * BUF_A + BUF_B) % 4096 == 0;
    static constexpr bool TRIPLE = NS8_TRIPLE != 0 && (FIT4 || FIT1);
    static constexpr int RT = TRIPLE ? (FIT4 ? 4 : 1) : R;           // interleaved copies of the WSiLU table
    static constexpr int OFF_B = BUF_A;
    static constexpr int OFF_A2 = BUF_A + BUF_B;
    static constexpr int OFF_TABLE = TRIPLE ? 2 * BUF_A + BUF_B : align16k(BUF_A + BUF_B);
    static constexpr int OFF_BIAS = OFF_TABLE + RT * TABLE_BYTES;
    static constexpr int OFF_Q = OFF_BIAS + BIAS_FLOATS * 4;
    // DW (round 6): the block's depthwise conv inside the launch. dc.0's output arrives as THREE runs of RH consecutive pixel rows
    // (the tile's pixels - 1 .. + 32 PXT of the picture rows above, at and below: pixels are row-major, a tile is a run of them) by
    // LDS-DMA into H, lane-linear; the nine taps [9][CI] and a line of zeros (the taps outside the picture) sit in front of it.
    // The conv of a tile is the work of NW = 8 or 4 waves, each with its own 32 PXT / NW pixels and ITS OWN pieces of H (whole
    // 1 KB DMA pieces from its first row on: the rows two neighbours both need are fetched twice): no wave waits for another's.
    static constexpr int run_chunks(int nw)
    {
        const int pxw = 32 * PXT / nw, pcs = ((pxw + 2) * (CI / 8) + 63) / 64;       // a wave's pixels, its DMA pieces per run
        return (nw - 1) * pxw * (CI / 8) + pcs * 64;
    }
    static constexpr int rows_per_run()
    {
        const int ch = run_chunks(8) > run_chunks(4) ? run_chunks(8) : run_chunks(4);
        const int r = (ch + CI / 8 - 1) / (CI / 8);
        return r > 32 * PXT + 2 ? r : 32 * PXT + 2;
    }
    static constexpr int RH = rows_per_run();                            // 68 / 36 rows of 128 channels, 36 of 192
    static constexpr int OFF_WDW = OFF_Q + 2 * C * 2;
    static constexpr int OFF_ZERO = OFF_WDW + 9 * CI * 2;
    static constexpr int OFF_H = (OFF_ZERO + 16 + 1023) & ~1023;
    static constexpr int BYTES = DW != 0 ? OFF_H + 3 * RH * CI * 2 : OFF_Q + 2 * C * 2;
    static_assert(BYTES <= 160 * 1024, "LDS budget");
    static_assert(DW == 0 || TRIPLE, "depthwise conv inside: needs the second t2 buffer");
};

// Per-wave tile / fragment counts. A "tile" = 32 output channels; a fragment = one MFMA "A" operand (32 channels x 16 k).
template <int C, int CI>
struct Geo {
    static constexpr int KS_C = C / 16, KS_I = CI / 16;
    static constexpr int QC = C / 128, QI = CI / 128;                  // tiles per SIMD pair of a C- / CI-wide layer
    static constexpr int HI_C = (QC + 1) / 2, LO_C = QC / 2;           // ... of which wave w < 4 / wave w + 4
    static constexpr int HI_I = (QI + 1) / 2, LO_I = QI / 2;
    // CI a multiple of 128: dc.0's tiles by SIMD pair as the C-wide layers'; otherwise (CI = 192) by the rule of the closing
    // convs below (waves 0 .. 3 one tile each, two of the waves 4 .. 7 one each)
    static constexpr bool I_BY_PAIR = CI % 128 == 0;
    // ffn.0: 4 CI / 32 tiles in pairs (a pair = 16 channels of t); the waves 0 .. 3 own P0_HI pairs each, 4 .. 7 P0_LO
    // (equal but for CI = 192: 2 | 1)
    static constexpr int PAIRS = CI / 16, P0_HI = (PAIRS + 7) / 8, P0_LO = (PAIRS - 4 * P0_HI) / 4;
    static constexpr int TP = 2;                                        // passes of 2 tiles
    static constexpr int n0(bool hiw) { return 2 * (hiw ? P0_HI : P0_LO); }      // ffn.0 tiles per wave
    static constexpr int np(bool hiw) { return hiw ? P0_HI : P0_LO; }
    // C a multiple of 128: the C-wide layers' tiles by SIMD pair; otherwise (C = 192) by the closing convs' rule - the waves
    // without a tile (6, 7) still walk a tile of ZERO weights through dc.3 / ffn.2 (one barrier sequence, one ring discipline)
    // and keep their results to themselves
    static constexpr bool C_BY_PAIR = C % 128 == 0;
    static constexpr bool EVEN = HI_C == LO_C && HI_I == LO_I && I_BY_PAIR && C_BY_PAIR && P0_HI == P0_LO;
    static constexpr int nt_c(bool hiw) { return C_BY_PAIR ? (hiw ? HI_C : LO_C) : (nt_f(C, hiw) > 0 ? nt_f(C, hiw) : 0); }
    static constexpr int nt_i(bool hiw) { return I_BY_PAIR ? (hiw ? HI_I : LO_I) : nt_f(CI, hiw); }
    static constexpr int f_dc3(bool hiw) { return nt_c(hiw) * KS_I; }
    static constexpr int f_ffn0(bool hiw) { return n0(hiw) * KS_C; }
    static constexpr int f_main(bool hiw) { return 2 * f_dc3(hiw) + f_ffn0(hiw); }
    static constexpr int f_dc0(bool hiw) { return nt_i(hiw) * KS_C; }
    // a chain-closing conv of width NN (C -> NN, NN / 32 tiles): the waves 0 .. 3 own nf_hi tiles each, of the waves 4 .. 7 the
    // first act_lo own nf_lo each (NN = 192: 1 | 1, two active; NN = 128: 1 | 0). Wave w < 4: tiles [w nf_hi, ...), wave w >= 4:
    // tiles [4 nf_hi + (w - 4) nf_lo, ...).
    static constexpr int nf_hi(int nn) { return (nn / 32 + 7) / 8; }
    static constexpr int nf_lo(int nn) { return nn / 32 - 4 * nf_hi(nn) <= 0 ? 0 : (nn / 32 - 4 * nf_hi(nn) + 3) / 4; }
    static constexpr int act_lo(int nn) { return nf_lo(nn) == 0 ? 0 : (nn / 32 - 4 * nf_hi(nn)) / nf_lo(nn); }
    static constexpr int nt_f(int nn, bool hiw) { return hiw ? nf_hi(nn) : nf_lo(nn); }
    static constexpr bool fin_ok(int nn) { return nn % 32 == 0 && nn >= 128 && 4 * nf_hi(nn) + act_lo(nn) * nf_lo(nn) == nn / 32; }
    // tiles / fragments of the NEXT slot (0: none, 1: dc.0 of the next block, NN > 1: closing conv of width NN)
    static constexpr int nt_next(int next, bool hiw) { return next == 0 ? 0 : next == 1 ? nt_i(hiw) : nt_f(next, hiw); }
    static constexpr int f_next(int next, bool hiw) { return nt_next(next, hiw) * KS_C; }
    static constexpr bool even(int next) { return EVEN && (next <= 1 || (nf_hi(next) == nf_lo(next) && act_lo(next) == 4)); }
    static_assert(C % 64 == 0 && CI % 64 == 0 && 4 * (P0_HI + P0_LO) == PAIRS && P0_LO >= 1, "channel counts the eight waves can share");
};

template <int C, int CI, int PXT, int NEXT, int DW, bool HIW>
__device__ __forceinline__ void block_body(const NsParams& p, char* const smem)
{
    using G = Geo<C, CI>;
    constexpr int RING = PXT == 1 ? NS8_RING1 : NS8_RING;           // (shadows the namespace's value: every use below is the body's)
    constexpr bool FIN = NEXT > 1;                                  // the NEXT slot is a chain-closing conv of width NEXT
    static_assert(!FIN || G::fin_ok(NEXT), "closing conv: a width the eight waves can share");
    static_assert(G::I_BY_PAIR || G::fin_ok(CI), "dc.0: an inner width the eight waves can share");
    static_assert(G::C_BY_PAIR || G::fin_ok(C), "a block width the eight waves can share");
    constexpr int PX = 32 * PXT;
    constexpr int KS_C = G::KS_C, KS_I = G::KS_I;
    constexpr int NT_C = G::nt_c(HIW), NP = G::np(HIW), TP = G::TP;
    constexpr int F_DC3 = G::f_dc3(HIW), F_FFN0 = G::f_ffn0(HIW), F_MAIN = G::f_main(HIW), F_DC0 = G::f_next(NEXT, HIW);
    constexpr int NT_N = G::nt_next(NEXT, HIW);                     // this wave's 32-channel tiles of the NEXT slot
    constexpr int CIP = Lay<C, CI, PXT, NEXT, DW>::CIP;            // LDS row of the CI-wide tensors (padded to 16-chunk groups)
    constexpr int CP = Lay<C, CI, PXT, NEXT, DW>::CP;
    constexpr int CH_C = CP / 8, CH_I = CIP / 8;                // 16-byte chunks per LDS row
    constexpr int PITCH_C = CP * 2, PITCH_I = CIP * 2;
    using L = Lay<C, CI, PXT, NEXT, DW>;
    constexpr bool TRIPLE = L::TRIPLE;
    constexpr int RT = L::RT;
    constexpr int OFF_B = L::OFF_B, OFF_A2 = L::OFF_A2;
    constexpr int OFF_TABLE = L::OFF_TABLE;
    constexpr int OFF_BIAS = L::OFF_BIAS;                           // fp32: b3 (C) | b0 (4 CI) | b2 (C) | b1n (CI; closing conv: NN)
    constexpr int BIAS_FLOATS = L::BIAS_FLOATS;
    constexpr int OFF_Q = L::OFF_Q;                                 // fp16: q | q2   (a closing conv's qf stays in memory: 16 bytes per 8 outputs)
    constexpr int TOTAL = F_MAIN + (NEXT ? F_DC0 : 0);
    static_assert((PX * CH_C) % NTHREADS == 0 && (PX * CH_I) % NTHREADS == 0, "tile rows must split evenly over the threads");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int simd = wave & 3;                            // the SIMD pair (waves simd, simd + 4)
    const int px = lane & 31;
    const int hi = lane >> 5;
    const int ntiles = (p.M + PX - 1) / PX;
    // persistent: tiles blockIdx.x, + gridDim.x, ... With the depthwise conv inside (DW) a tile also reads dc.0's output of the picture
    // rows above and below it - tiles W / (32 PXT) further up and down: there the tiles go to the workgroups in eight BANDS, one per XCD
    // (workgroup w runs on XCD w mod 8), so that those rows are in the XCD's own L2 (fetched by a neighbour) instead of in memory
    const bool banded = DW != 0 && (gridDim.x & 7) == 0 && ntiles >= static_cast<int>(gridDim.x);
    const int band0 = banded ? static_cast<int>((blockIdx.x & 7) * ntiles) >> 3 : 0;
    const int band_n = banded ? (static_cast<int>(((blockIdx.x & 7) + 1) * ntiles) >> 3) - band0 : 0;
    int slot = blockIdx.x >> 3;                          // banded: this tile's place in the band (band_n >= gridDim.x / 8: the first one exists)
    int tile = banded ? band0 + slot : static_cast<int>(blockIdx.x);
    bool first_tile = true;
    int m0 = tile * PX;
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) void*)smem));
    if ((lds_base & 16383u) != 0) __builtin_trap();      // dynamic LDS starts at 0 (no static LDS in this kernel)
    // first channel of this wave's share of a C-wide / CI-wide layer, of ffn.0's 4 CI channels, of t
    // (HIW = the body of the waves with the larger share; where the shares are equal every wave runs it)
    const bool upper = wave >= 4;
    const int cb_c = G::C_BY_PAIR ? 32 * (simd * G::QC + (upper ? G::HI_C : 0))
                   : HIW ? 32 * wave * G::nf_hi(C) : 32 * (4 * G::nf_hi(C) + (wave - 4) * G::nf_lo(C));
    // (C = 192: the waves 6 and 7 own no tile of the C-wide layers: their "tile" lies in the padding of the LDS rows, its weights are
    // zeros, nothing of it is loaded from or stored to memory)
    const bool c_on = G::C_BY_PAIR || HIW || (wave - 4) < G::act_lo(C);
    const int cb_x = c_on ? cb_c : 0;
    const int cb_0 = HIW ? wave * (32 * G::n0(true)) : 4 * 32 * G::n0(true) + (wave - 4) * (32 * G::n0(false));
    // the NEXT slot: first channel of this wave's share, and whether it has one (closing conv of 192 channels: waves 6, 7 idle)
    constexpr bool BY_PAIR = !FIN && G::I_BY_PAIR;                   // dc.0 of a block whose inner width is a multiple of 128
    constexpr int NN = FIN ? NEXT : CI;                              // width of the NEXT slot's output
    const int cb_n = BY_PAIR ? 32 * (simd * G::QI + (upper ? G::HI_I : 0))
                   : HIW ? 32 * wave * G::nf_hi(NN) : 32 * (4 * G::nf_hi(NN) + (wave - 4) * G::nf_lo(NN));
    const bool nx_on = BY_PAIR || HIW || (wave - 4) < G::act_lo(NN);
    // stamps (wave 0, first tile): 0 entry | 1 constants + first t2 in LDS | 2 dc.3 done (barrier) | one per ffn.0 pass |
    // ffn.0 done (barrier) | ffn.2 MFMAs | ffn.2 done | dc.0 MFMAs | dc.0 done
    const long long rt0 = static_cast<long long>(__builtin_amdgcn_s_memrealtime());
    int stamp_no = 0;
    auto stamp = [&]() {
        if (p.timeline != nullptr && tid == 0 && stamp_no < 29 && first_tile) {
            p.timeline[static_cast<size_t>(blockIdx.x) * 32 + stamp_no] = static_cast<long long>(__builtin_readcyclecounter());
        }
        ++stamp_no;
    };
    stamp();

    // ---- constants into registers first (all loads independent), written to LDS behind the first tile's transfers
    // RT copies of the 256 table rows, interleaved: slot s = row s / RT (RT = 4: two slots per thread; RT = 1: the first 256 threads)
    static_assert(RT == 4 || RT == 1, "table slots per thread, spelled out");
    const float4 tab0 = p.wsilu[min(tid / RT, WSILU_SEGMENTS - 1)], tab1 = p.wsilu[min((tid + NTHREADS) / RT, WSILU_SEGMENTS - 1)];
    constexpr int CONST_UNITS = (BIAS_FLOATS + 2 * C) / 8;      // b3 | b0 | b2 | b1n | q | q2 in 8-channel units
    constexpr int CONST_PER_THREAD = (CONST_UNITS + NTHREADS - 1) / NTHREADS;
    static_assert(CONST_PER_THREAD <= 2, "two named registers below");
    auto const_unit = [&](int k) {
        const int ch = min(tid + k * NTHREADS, CONST_UNITS - 1) * 8;
        const half_t* src = ch < C ? p.b3 + ch
                          : ch < C + 4 * CI ? p.b0 + (ch - C)
                          : ch < 2 * C + 4 * CI ? p.b2 + (ch - C - 4 * CI)
                          : ch < BIAS_FLOATS ? (p.b1n != nullptr ? p.b1n + (ch - 2 * C - 4 * CI) : p.b2)
                          : ch < BIAS_FLOATS + C ? (p.q != nullptr ? p.q + (ch - BIAS_FLOATS) : p.b2)
                          : (p.q2 != nullptr ? p.q2 + (ch - BIAS_FLOATS - C) : p.b2);
        return *reinterpret_cast<const half8*>(src);
    };
    const half8 cv0 = const_unit(0), cv1 = const_unit(CONST_PER_THREAD > 1 ? 1 : 0);
    static_assert(DW == 0 || 9 * CI / 8 <= NTHREADS, "the depthwise taps: one 16-byte unit per thread");
    half8 wdv = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (DW != 0) wdv = *reinterpret_cast<const half8*>(p.wdw + min(tid, 9 * CI / 8 - 1) * 8);
    const float* const lb3 = reinterpret_cast<const float*>(smem + OFF_BIAS);
    const float* const lb0 = lb3 + C;
    const float* const lb2 = lb3 + C + 4 * CI;
    const float* const lb1n = lb3 + 2 * C + 4 * CI;
    const half_t* const lq = reinterpret_cast<const half_t*>(smem + OFF_Q);
    const half_t* const lq2 = lq + C;
    unsigned tab = lds_base + OFF_TABLE + (lane & (RT - 1)) * 16;

    // ---- the wave's weight streams (dcb_nsplit.hip pack_*8): waves 0 .. 3 first (their share may be the larger one),
    // then 4 .. 7; fragment f of a wave at 1 KB f from the wave's base, lane-linear. Buffer loads: lane offset + 12-bit
    // immediate + one scalar offset per 4 KB, no VALU per fragment.
    constexpr int FM_HI = G::f_main(true), FM_LO = G::f_main(false), FD_HI = G::f_next(NEXT, true), FD_LO = G::f_next(NEXT, false);
    unsigned wsm = static_cast<unsigned>((HIW ? wave * FM_HI : 4 * FM_HI + (wave - 4) * FM_LO) * 64 + lane) * 16u;
    // (a wave without a share of the closing conv still prefetches ring-deep behind ffn.2: from the stream's first bytes)
    unsigned wsn = static_cast<unsigned>((HIW ? wave * FD_HI : nx_on ? 4 * FD_HI + (wave - 4) * FD_LO : 0) * 64 + lane) * 16u;
    const __amdgpu_buffer_rsrc_t rs_main = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8*>(p.wmain), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(const_cast<half8*>(NEXT != 0 ? p.wnext : p.wmain), 0, 0x7fffffff, 0x00020000);
    half8 ring[RING];
    auto issue = [&](auto f_tag) {
        constexpr int f = decltype(f_tag)::value;
        if constexpr (f < F_MAIN) {
            ring[f % RING] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_main, wsm + static_cast<unsigned>(f & 3) * 1024u, (f >> 2) * 4096, 0));
        } else if constexpr (f < TOTAL) {
            constexpr int g = f - F_MAIN;
            ring[f % RING] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs_next, wsn + static_cast<unsigned>(g & 3) * 1024u, (g >> 2) * 4096, 0));
        }
    };
    // ---- a [PX][W] tile of whole rows, memory -> LDS by LDS-DMA; LDS image lane-linear, the bank swizzle (16-byte
    // chunk c of row r lives at chunk c ^ (r & 15)) sits on the SOURCE side. Rows behind the picture read its last row.
    int tidv = tid;
    auto dma_tile = [&](auto chunks_tag, const half_t* base, int ld, unsigned lds_off, int first_row) {
        constexpr int CHN = decltype(chunks_tag)::value;
        const int f = min(first_row, p.M - 1);
        const int last = p.M - 1 - f;
        const half_t* const w = base + static_cast<size_t>(f) * ld;
#pragma unroll
        for (int i = 0; i < PX * CHN / NTHREADS; ++i) {
            const int pos = i * NTHREADS + tidv;
            const int r = pos / CHN, pc = pos % CHN;
            int lc = pc ^ (r & 15);
            if constexpr (CHN * 8 != CI) lc = min(lc, CI / 8 - 1);      // a slot of the padding: any chunk of the row will do
            const int rr = min(r, last);
            lds_dma16(w, static_cast<unsigned>(rr * ld + lc * 8) * 2u, lds_base + lds_off + (i * NTHREADS + wave * 64) * 16);
        }
    };
    using ChI = std::integral_constant<int, CH_I>;
    // ---- DW: dc.0's output around the pixels of wave `wi` of NW -> H (three runs of RH pixel rows from pixel first_row - 1 of the
    // picture rows above / at / below; rows clamped to the tensor - what lies outside the picture is never read back)
    constexpr int RH = L::RH;
    constexpr int CH_H = CI / 8;                                    // 16-byte chunks per row of H (unpadded, unswizzled)
    auto dma_halo = [&](auto nw_tag, int wi, int first_row) {
        constexpr int NW = decltype(nw_tag)::value;
        constexpr int PXW = PX / NW, PCS = ((PXW + 2) * CH_H + 63) / 64;
        static_assert((NW - 1) * PXW * CH_H + PCS * 64 <= RH * CH_H, "a wave's pieces stay inside the run");
        const int r0 = __builtin_amdgcn_readfirstlane(first_row) - 1 + wi * PXW;      // the first row of this wave's pieces of the middle run
        if (64 % CH_H == 0 && r0 - p.W >= 0 && r0 + p.W + PCS * (64 / CH_H) <= p.M) {
            // whole rows per piece and none of them clamped: one lane offset, the pieces' addresses in scalar registers
            const unsigned lane_off = static_cast<unsigned>(((tidv & 63) / CH_H) * p.ldt + ((tidv & 63) % CH_H) * 8) * 2u;
#pragma unroll
            for (int run = 0; run < 3; ++run)
#pragma unroll
                for (int i = 0; i < PCS; ++i) {
                    const half_t* const src = p.t1 + static_cast<size_t>(r0 + (run - 1) * p.W + i * (64 / CH_H)) * p.ldt;
                    lds_dma16(src, lane_off, lds_base + L::OFF_H + (run * (RH * CH_H) + wi * (PXW * CH_H) + i * 64) * 16);
                }
            return;
        }
#pragma unroll
        for (int run = 0; run < 3; ++run)
#pragma unroll
            for (int i = 0; i < PCS; ++i) {
                const int pos = wi * (PXW * CH_H) + i * 64 + (tidv & 63);     // chunk of the run
                const int j = pos / CH_H, pc = pos % CH_H;
                const int g = min(max(first_row + (run - 1) * p.W - 1 + j, 0), p.M - 1);
                lds_dma16(p.t1, static_cast<unsigned>(g * p.ldt + pc * 8) * 2u, lds_base + L::OFF_H + (run * (RH * CH_H) + wi * (PXW * CH_H) + i * 64) * 16);
            }
    };
    constexpr int H_PER_WAVE8 = 3 * (((PX / 8 + 2) * CH_H + 63) / 64);      // pieces per wave when all eight share a tile
    // ---- DW: the depthwise 3x3 conv of a tile, H -> the t2 buffer at `dst` (swizzled as dma_tile leaves it), by NW waves of which
    // this one is number `wi`: fp32 fmaf chain over the taps (ky, kx) ascending from 0, taps outside the picture read zeros
    // (dwconv.hip skips them: the same sum - a chain that starts at + 0 never reaches - 0)
    auto dw_tile = [&](auto nw_tag, int wi, int dst, int first_row) {
        constexpr int NW = decltype(nw_tag)::value;
        constexpr int PXW = PX / NW;                                 // this wave's pixels: wi PXW .. + PXW - 1 of the tile
        constexpr int ITEMS = PXW * CH_H;                            // its (pixel, 8-channel chunk) pairs, 64 per step
        constexpr int STEPS = (ITEMS + 63) / 64;
        constexpr bool FIXED_CHUNK = 64 % CH_H == 0;                 // a lane keeps its chunk from step to step: the taps stay in registers
        half8 wt[9];
        if constexpr (FIXED_CHUNK) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const half8*>(smem + L::OFF_WDW + (k * CI + (tidv & (CH_H - 1)) * 8) * 2);
        }
        // the tile's first pixel in the picture (wave-uniform: scalar registers)
        const int fr = __builtin_amdgcn_readfirstlane(first_row);
        const int y0 = fr / p.W, x0 = fr - y0 * p.W;
        const int n0 = wi * ITEMS;                                   // H is [3][RH][CH_H] chunks: tap (ky, kx) of item n at chunk n0 + n + (ky RH + kx) CH_H
        // the nine taps of step i's item into registers
        auto taps = [&](int i, half8 (&v)[9]) {
            const int n = i * 64 + (tidv & 63);
            const int base = L::OFF_H + (n0 + n) * 16;
            // the step's pixels r_lo .. r_hi of the tile (scalar): all of them inside the picture with all their neighbours?
            const int r_lo = wi * PXW + i * 64 / CH_H, r_hi = wi * PXW + (i * 64 + 63) / CH_H;
            int xf = x0 + r_lo, yf = y0;
            while (xf >= p.W) { xf -= p.W; ++yf; }
            const bool inner = yf > 0 && fr + r_hi + p.W < p.M && xf >= 1 && xf + (r_hi - r_lo) + 1 < p.W;
            if (inner) {
#pragma unroll
                for (int k = 0; k < 9; ++k) v[k] = *reinterpret_cast<const half8*>(smem + base + ((k / 3) * RH + k % 3) * CH_H * 16);
            } else {
                const int r = wi * PXW + n / CH_H;
                const int m = first_row + r;
                int xx = xf + (r - r_lo), yy = yf;
                while (xx >= p.W) { xx -= p.W; ++yy; }
                const bool up = yy > 0, down = m + p.W < p.M, left = xx > 0, right = xx + 1 < p.W;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int ky = k / 3, kx = k % 3;
                    const bool ok = (ky != 0 || up) && (ky != 2 || down) && (kx != 0 || left) && (kx != 2 || right);
                    v[k] = *reinterpret_cast<const half8*>(smem + (ok ? base + (ky * RH + kx) * CH_H * 16 : L::OFF_ZERO));
                }
            }
        };
        // (the taps of step i + 1 are read while step i's chain runs: a read-use pair per tap, as the compiler lays the loop out by
        // itself, exposes the LDS latency nine times per item)
        half8 v[2][9];
        taps(0, v[0]);
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            if (i + 1 < STEPS) taps(i + 1, v[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int n = i * 64 + (tidv & 63);
            const int r = wi * PXW + n / CH_H, pc = n % CH_H;        // pixel of the tile, chunk of its row
            if constexpr (!FIXED_CHUNK) {
#pragma unroll
                for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const half8*>(smem + L::OFF_WDW + (k * CI + pc * 8) * 2);
            }
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(static_cast<float>(v[i & 1][k][e]), static_cast<float>(wt[k][e]), acc[e]);
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = to_half(acc[e]);
            if (ITEMS % 64 == 0 || n < ITEMS) *reinterpret_cast<half8*>(smem + dst + r * PITCH_I + ((pc ^ (r & 15)) << 4)) = o;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // (where the waves of a SIMD pair run different bodies, the depthwise conv of the NEXT tile is the work of the waves 4 .. 7: their
    // share of the NEXT slot is the smaller one - none at all for dc.0 of a (256, 128) block)
    constexpr bool DW_ALL = G::even(NEXT);
    // ---- the block input x of a tile (dc.3's residual) waits in registers, in the layout of dc.3's epilogue
    half8 xr[NT_C > 0 ? NT_C : 1][PXT][2];
    int pxv = px, hiv = hi;
    auto load_x = [&](int first_row) {
#pragma unroll
        for (int t = 0; t < PXT; ++t) {
            const half_t* const row = p.x + static_cast<size_t>(min(first_row + 32 * t + pxv, p.M - 1)) * p.ldx + (cb_x + 8 * hiv);
#pragma unroll
            for (int j = 0; j < NT_C; ++j)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) xr[j][t][pr] = *reinterpret_cast<const half8*>(row + 32 * j + 16 * pr);
        }
    };
    // first tile: t2 -> A, the first weight fragments, x; then the constants go to LDS
    if constexpr (DW != 0) dma_halo(std::integral_constant<int, 8>{}, wave, m0);
    else dma_tile(ChI{}, p.t2, p.ldt, 0, m0);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, RING>([&](auto i) { issue(i); });
    __builtin_amdgcn_sched_barrier(0);
    load_x(m0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DW != 0 ? H_PER_WAVE8 : PX * CH_I / NTHREADS) + RING + NT_C * PXT * 2) : "memory");     // the constants have arrived
    {
        float4* t = reinterpret_cast<float4*>(smem + OFF_TABLE);
        if (RT == 4 || tid < WSILU_SEGMENTS) t[tid] = tab0;
        if (RT == 4) t[tid + NTHREADS] = tab1;
        float* lb = reinterpret_cast<float*>(smem + OFF_BIAS);
        half_t* lqw = reinterpret_cast<half_t*>(smem + OFF_Q);
        auto put = [&](int k, const half8 v) {
            const int u = tid + k * NTHREADS;
            if (u < BIAS_FLOATS / 8) {
                float4v lo4, hi4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo4[e] = static_cast<float>(v[e]);
                    hi4[e] = static_cast<float>(v[4 + e]);
                }
                *reinterpret_cast<float4v*>(lb + u * 8) = lo4;
                *reinterpret_cast<float4v*>(lb + u * 8 + 4) = hi4;
            } else if (u < CONST_UNITS) {
                *reinterpret_cast<half8*>(lqw + (u - BIAS_FLOATS / 8) * 8) = v;
            }
        };
        put(0, cv0);
        if (CONST_PER_THREAD > 1) put(1, cv1);
        if constexpr (DW != 0) {
            if (tid < 9 * CI / 8) *reinterpret_cast<half8*>(smem + L::OFF_WDW + tid * 16) = wdv;
            if (tid == NTHREADS - 1) *reinterpret_cast<half8*>(smem + L::OFF_ZERO) = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // this wave's pieces of t2 have landed (x, requested last, may still be on its way: dc.3's epilogue is its first use)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT_C * PXT * 2) : "memory");
    __syncthreads();
    if constexpr (DW != 0) {
        // the first tile's depthwise conv: all eight waves, H -> A
        dw_tile(std::integral_constant<int, 8>{}, wave, 0, m0);
        __syncthreads();
    }
    stamp();

    // ---- fragment addressing. Row of pixel tile t: (32 t + px) * pitch; chunk c of a row sits at c ^ (px & 15).
    // B fragment of k-slice ks: chunk 2 ks + hi = (2 ks) ^ hi: (32 ks) ^ s0 = ((32 (ks & 7)) ^ s0) + 256 (ks >> 3).
    int s0 = (hi ^ (px & 15)) << 4;
    int rowA = px * PITCH_I, rowB = px * PITCH_C + OFF_B;     // byte offsets from smem (TRIPLE: rowA of the tile's own t2 buffer)
    int abuf = 0;                                             // TRIPLE: byte offset of the buffer this tile's t2 / t live in (0 or OFF_A2)
    int hi4 = 4 * hi;
    int fa8[8], fb8[8];
    auto frag_a = [&](int t, int ks) { return *reinterpret_cast<const half8*>(smem + fa8[ks & 7] + (t * (32 * PITCH_I) + (ks >> 3) * 256)); };
    auto frag_b = [&](int t, int ks) { return *reinterpret_cast<const half8*>(smem + fb8[ks & 7] + (t * (32 * PITCH_C) + (ks >> 3) * 256)); };
    // the 16-byte run of channels ch0 + 8 hi .. + 7 (ch0 a multiple of 16) of this lane's pixel in tile t
    auto run_a = [&](int t, int ch0) { return reinterpret_cast<half8*>(smem + rowA + t * (32 * PITCH_I) + ((ch0 * 2) ^ s0)); };
    auto run_b = [&](int t, int ch0) { return reinterpret_cast<half8*>(smem + rowB + t * (32 * PITCH_C) + ((ch0 * 2) ^ s0)); };
    // accumulator tile (32 channels from `first`) initialised with the bias: acc[r] = channel first + 8 (r>>2) + 4 hi + (r&3)
    auto bias_tile = [&](float16v& acc, const float* bias, int first) {
        const float* bp = bias + first + hi4;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4v b4 = *reinterpret_cast<const float4v*>(bp + 8 * g4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g4 + e] = b4[e];
        }
    };
    // accumulator tile -> run pr: channels 16 pr + 8 hi .. + 7 of the tile, this lane's pixel (half-waves paired up)
    auto runs_of = [&](const float16v& a, int pr, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * pr + e]), __float_as_uint(a[8 * pr + 4 + e]), false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
        }
    };
    // NT tiles x PXT pixel tiles over KSN k-slices (fragment F0 + ks NT + j), activations through `frag` (A or B)
    auto contract = [&](auto nt_tag, auto ks_tag, auto f0_tag, auto&& frag, auto& acc) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr int KSN = decltype(ks_tag)::value;
        constexpr int F0 = decltype(f0_tag)::value;
        // activation fragments, read AH k-slices ahead of their MFMAs (NS8_AHEAD; one slice is 32 NT PXT matrix-core cycles:
        // with one tile per wave that does not cover a ds_read_b128 under load - round 6, profiles/r06_ahead.txt)
        constexpr int AH = NT * PXT >= 4 ? NS8_AHEAD_WIDE : NS8_AHEAD;
        half8 b[AH + 1][PXT];
        static_for<0, AH>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            if constexpr (i < KSN) {
#pragma unroll
                for (int t = 0; t < PXT; ++t) b[i][t] = frag(t, i);
            }
        });
        static_for<0, KSN>([&](auto kt) {
            constexpr int ks = decltype(kt)::value;
            if constexpr (ks + AH < KSN) {
#pragma unroll
                for (int t = 0; t < PXT; ++t) b[(ks + AH) % (AH + 1)][t] = frag(t, ks + AH);
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NT>([&](auto j_tag) {
                constexpr int j = NT - 1 - decltype(j_tag)::value;       // the slice's LAST fragment first: one counted wait per slice
                const half8 a = ring[(F0 + ks * NT + j) % RING];
#pragma unroll
                for (int t = 0; t < PXT; ++t) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[ks % (AH + 1)][t], acc[j][t], 0, 0, 0);
            });
            static_for<0, NT>([&](auto j_tag) { issue(std::integral_constant<int, F0 + ks * NT + decltype(j_tag)::value + RING>{}); });
            // nothing crosses a k-slice (left alone, hipcc sinks every prefetch load down to the MFMA that consumes it)
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using KsC = std::integral_constant<int, KS_C>;
    using KsI = std::integral_constant<int, KS_I>;

    // ================================================================ persistent loop over this workgroup's tiles
    for (;;) {
    // (made opaque once per tile: as loop invariants every derived address would be hoisted out of the loop and spilled)
    asm volatile("" : "+v"(wsm), "+v"(wsn), "+v"(tab), "+v"(s0), "+v"(rowA), "+v"(rowB), "+v"(hi4), "+v"(tidv), "+v"(pxv), "+v"(hiv));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa8[i] = rowA + ((32 * i) ^ s0);
        fb8[i] = rowB + ((32 * i) ^ s0);
    }
    const int next_slot = slot + static_cast<int>(gridDim.x >> 3);
    const int next_tile = banded ? band0 + next_slot : tile + static_cast<int>(gridDim.x);
    const bool has_next = banded ? next_slot < band_n : next_tile < ntiles;
    // ================================================================ dc.3: y1 = W3 t2 + b3' + x   (A -> B)
    if constexpr (NT_C > 0) {
        float16v acc[NT_C][PXT];
#pragma unroll
        for (int j = 0; j < NT_C; ++j)
#pragma unroll
            for (int t = 0; t < PXT; ++t) bias_tile(acc[j][t], lb3, cb_c + 32 * j);
        contract(std::integral_constant<int, NT_C>{}, KsI{}, std::integral_constant<int, 0>{}, frag_a, acc);
#pragma unroll
        for (int j = 0; j < NT_C; ++j)
#pragma unroll
            for (int t = 0; t < PXT; ++t)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float v[8];
                    runs_of(acc[j][t], pr, v);
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = to_half(v[e] + static_cast<float>(xr[j][t][pr][e]));
                    *run_b(t, cb_c + 32 * j + 16 * pr) = o;
                }
    }
    __syncthreads();            // y1 complete in B; every wave is done with t2 in A
    stamp();

    // ================================================================ ffn.0: t = chunk_add(WSiLU(W0 y1 + b0))   (B -> A)
    // The wave's 32 N0 ffn.0 channels in NP passes of TP = 2 tiles: a pass = 64 ffn.0 channels = 16 channels of t
    // (the lower half-wave collects the 8 of the pass's first tile, the upper half-wave those of the second).
    {
        float16v acc[TP][PXT];
        static_for<0, NP>([&](auto pass_tag) {
            constexpr int pass = decltype(pass_tag)::value;
            const int f0 = cb_0 + pass * (32 * TP);        // first ffn.0 channel of the pass
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int t = 0; t < PXT; ++t) bias_tile(acc[j][t], lb0, f0 + 32 * j);
            contract(std::integral_constant<int, TP>{}, KsC{}, std::integral_constant<int, F_DC3 + pass * TP * KS_C>{}, frag_b, acc);
            if constexpr (TRIPLE && pass == NP - 1) {
                // the next tile's t2 into the OTHER buffer and its x into registers, in front of an epilogue that needs no
                // weights (the ring already holds ffn.2's first fragments: they return before these transfers)
                if constexpr (DW != 0) {
                    // (H is free: this tile's depthwise conv ran during the previous tile. Where the waves 4 .. 7 do the conv, they alone
                    // fetch: behind these transfers a wave's weight fragments would wait - loads return in order - and the waves 0 .. 3
                    // have the NEXT slot's still to come)
                    if (has_next) {
                        if constexpr (DW_ALL) dma_halo(std::integral_constant<int, 8>{}, wave, next_tile * PX);
                        else if constexpr (!HIW) dma_halo(std::integral_constant<int, 4>{}, wave - 4, next_tile * PX);
                    }
                } else {
                    if (has_next) dma_tile(ChI{}, p.t2, p.ldt, abuf == 0 ? OFF_A2 : 0, next_tile * PX);
                }
                if constexpr (NT_C > 0) load_x(next_tile * PX);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < PXT; ++t) {
                float sums[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int g0 = 0; g0 < 16; g0 += GB) {
                        float4v c[GB];
#pragma unroll
                        for (int e = 0; e < GB; ++e) {
                            const float4 r = wsilu_row_lds<RT, true>(acc[h][t][g0 + e], tab);
                            c[e] = float4v{r.x, r.y, r.z, r.w};
                        }
#pragma unroll
                        for (int g = 0; g < GB / 4; ++g) {
                            auto row = [&](int e) { const float4v r = c[4 * g + e]; return make_float4(r[0], r[1], r[2], r[3]); };
                            const int v0 = g0 + 4 * g;
                            float s = acc[h][t][v0] * wsilu_poly(acc[h][t][v0], row(0));
#pragma unroll
                            for (int e = 1; e < 4; ++e) s = fmaf(acc[h][t][v0 + e], wsilu_poly(acc[h][t][v0 + e], row(e)), s);
                            sums[h][g0 / 4 + g] = s;
                        }
                    }
                }
                half8 o;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sums[0][g]), __float_as_uint(sums[1][g]), false, false);
                    o[2 * g] = to_half(__uint_as_float(sw[0]));
                    o[2 * g + 1] = to_half(__uint_as_float(sw[1]));
                }
                *run_a(t, f0 / 4) = o;                 // t channels f0 / 4 + 8 hi .. + 7
            }
            stamp();
        });
    }
    __syncthreads();            // t complete in A; every wave is done with y1 as an operand
    stamp();

    // ================================================================ ffn.2: y = (W2 t + b2 + y1 [+ x]) [* q] -> fp16 [* q2]   (A -> B in place of y1)
    {
        float16v acc[NT_C > 0 ? NT_C : 1][PXT];
        if constexpr (NT_C > 0) {
#pragma unroll
            for (int j = 0; j < NT_C; ++j)
#pragma unroll
                for (int t = 0; t < PXT; ++t) bias_tile(acc[j][t], lb2, cb_c + 32 * j);
            contract(std::integral_constant<int, NT_C>{}, KsI{}, std::integral_constant<int, F_DC3 + F_FFN0>{}, frag_a, acc);
        }
        stamp();
        // the next tile's transfers: t is dead once every wave is behind its last ffn.2 MFMA. They go out in front of
        // the epilogue, which needs no weights (a transfer from memory in front of a contraction holds up every weight
        // fragment requested behind it: loads return in order)
        if constexpr (!TRIPLE) {
            __syncthreads();
            if (has_next) dma_tile(ChI{}, p.t2, p.ldt, 0, next_tile * PX);
            if constexpr (NT_C > 0) load_x(next_tile * PX);        // (unconditional - rows are clamped to the picture)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NT_C; ++j)
#pragma unroll
            for (int t = 0; t < PXT; ++t)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int ch = cb_c + 32 * j + 16 * pr;          // + 8 hi
                    float v[8];
                    runs_of(acc[j][t], pr, v);
                    half8* const slot = run_b(t, ch);
                    const half8 y1 = *slot;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(y1[e]);
                    if (p.shortcut) {
                        const int m = min(m0 + 32 * t + px, p.M - 1);
                        const half8 r8 = *reinterpret_cast<const half8*>(p.x + static_cast<size_t>(m) * p.ldx + (cb_x + 32 * j + 16 * pr) + 8 * hi);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] + static_cast<float>(r8[e]);
                    }
                    if (p.q != nullptr) {
                        const half8 q8 = *reinterpret_cast<const half8*>(lq + ch + 8 * hi);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] * static_cast<float>(q8[e]);
                    }
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
                    if (p.q2 != nullptr) {
                        const half8 q8 = *reinterpret_cast<const half8*>(lq2 + ch + 8 * hi);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = hmul(o[e], q8[e]);
                    }
                    if constexpr (NEXT != 0) *slot = o;     // the NEXT slot's operand
                    const int m = m0 + 32 * t + pxv;
                    // (a block whose output only feeds the closing conv of its chain keeps it in LDS: y = null)
                    if (m < p.M && c_on && (!FIN || p.y != nullptr)) store_line(p.y + static_cast<size_t>(m) * p.ldy + ch + 8 * hiv, o);
                }
    }
    if constexpr (NEXT != 0) __syncthreads();       // y complete in B
    stamp();
    // ================================================================ the NEXT slot   (B -> memory)
    //   NEXT = 1: dc.0 of the next block, t1' = WSiLU(W1' y + b1');  NEXT = NN: the chain's closing conv, o = (Wf y + bf) [* qf]
    if constexpr (NEXT != 0 && NT_N > 0) {
        if (nx_on) {
            float16v acc[NT_N][PXT];
#pragma unroll
            for (int j = 0; j < NT_N; ++j)
#pragma unroll
                for (int t = 0; t < PXT; ++t) bias_tile(acc[j][t], lb1n, cb_n + 32 * j);
            contract(std::integral_constant<int, NT_N>{}, KsC{}, std::integral_constant<int, F_MAIN>{}, frag_b, acc);
            // the ring is empty: the first fragments of the next tile go out now and arrive under the epilogue below
            if (has_next) {
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, RING>([&](auto i) { issue(i); });
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp();
#pragma unroll
            for (int j = 0; j < NT_N; ++j)
#pragma unroll
                for (int t = 0; t < PXT; ++t)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float v[8];
                        runs_of(acc[j][t], pr, v);
                        half8 o;
                        if constexpr (!FIN) {
                            float4v c[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float4 r = wsilu_row_lds<RT, true>(v[e], tab);
                                c[e] = float4v{r.x, r.y, r.z, r.w};
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = to_half(v[e] * wsilu_poly(v[e], make_float4(c[e][0], c[e][1], c[e][2], c[e][3])));
                        } else {
                            if (p.qf != nullptr) {           // conv1x1_bias_with_quant: (acc) * q, one rounding
                                const half8 q8 = *reinterpret_cast<const half8*>(p.qf + cb_n + 32 * j + 16 * pr + 8 * hiv);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = v[e] * static_cast<float>(q8[e]);
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = to_half(v[e]);
                        }
                        const int m = m0 + 32 * t + pxv;
                        if (m < p.M) store_line(p.t1n + static_cast<size_t>(m) * p.ldt1 + cb_n + 32 * j + 16 * pr + 8 * hiv, o);
                    }
        } else {
            if (has_next) {
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, RING>([&](auto i) { issue(i); });
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp();
        }
    } else {
        if (has_next) {
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, RING>([&](auto i) { issue(i); });
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp();
    }
    if constexpr (DW != 0) {
        // the NEXT tile's depthwise conv, H -> the other t2 buffer: beside the NEXT slot (the waves 4 .. 7, whose share of it is the
        // smaller one - none at all for dc.0 of a (256, 128) block) or behind it (all waves). A wave reads only the pieces of H it
        // fetched itself: it waits for those (the ring's fragments of the next tile, requested later, may stay on their way)
        if (has_next) {
            if constexpr (DW_ALL || !HIW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING) : "memory");
            if constexpr (DW_ALL) dw_tile(std::integral_constant<int, 8>{}, wave, abuf == 0 ? OFF_A2 : 0, next_tile * PX);
            else if constexpr (!HIW) dw_tile(std::integral_constant<int, 4>{}, wave - 4, abuf == 0 ? OFF_A2 : 0, next_tile * PX);
        }
    }
    stamp();
    if (!has_next) break;
    // the next tile's t2 has landed (every wave waits for its own pieces, the barrier covers the others'); every wave is
    // done with y in B: B is free for the next tile's y1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp();        // (the first tile's last stamp: every wave has arrived - with the depthwise conv inside, the waves 4 .. 7 from theirs)
    tile = next_tile;
    slot = next_slot;
    first_tile = false;
    m0 = tile * PX;
    if constexpr (TRIPLE) {
        abuf = abuf == 0 ? OFF_A2 : 0;
        rowA = pxv * PITCH_I + abuf;
    }
    }       // tiles
    // stamp 31: the workgroup's last instruction (all tiles): shader cycles of the whole launch per workgroup; stamps 29 / 30: the
    // constant 100 MHz clock (s_memrealtime) at entry / here: cycles / time = the shader clock the launch really ran at
    if (p.timeline != nullptr && tid == 0) {
        p.timeline[static_cast<size_t>(blockIdx.x) * 32 + 31] = static_cast<long long>(__builtin_readcyclecounter());
        p.timeline[static_cast<size_t>(blockIdx.x) * 32 + 30] = static_cast<long long>(__builtin_amdgcn_s_memrealtime());
        p.timeline[static_cast<size_t>(blockIdx.x) * 32 + 29] = rt0;
    }
}

// NS8_OCC2 (experiment, round 6): 32-pixel workgroups whose LDS image fits twice into a CU are compiled for 128 registers, so that
// TWO workgroups share a CU (four waves per SIMD): the barrier / latency phases of one run under the contractions of the other
#ifndef NS8_OCC2
#define NS8_OCC2 0
#endif
template <int C, int CI, int PXT, int NEXT, int DW = 0>
constexpr int waves_per_simd() { return (NS8_OCC2 != 0 && PXT == 1 && Lay<C, CI, PXT, NEXT, DW>::BYTES <= 80 * 1024) ? 4 : 2; }

// DW = 1: the block's depthwise conv inside (p.t1 / p.wdw / p.W instead of p.t2)
template <int C, int CI, int PXT, int NEXT, int DW = 0>
__global__ void __launch_bounds__(NTHREADS, (waves_per_simd<C, CI, PXT, NEXT, DW>()))
dcb_nsplit8_kernel(const NsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem8[];
    if constexpr (Geo<C, CI>::even(NEXT)) {
        block_body<C, CI, PXT, NEXT, DW, true>(p, smem8);
    } else {
        // the waves of a SIMD pair own different numbers of tiles: two bodies, one barrier sequence
        if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) block_body<C, CI, PXT, NEXT, DW, true>(p, smem8);
        else block_body<C, CI, PXT, NEXT, DW, false>(p, smem8);
    }
}

template <int C, int CI, int PXT, int NEXT, int DW = 0>
void launch8(const NsParams& p, hipStream_t stream)
{
    auto kern = dcb_nsplit8_kernel<C, CI, PXT, NEXT, DW>;
    constexpr int smem = Lay<C, CI, PXT, NEXT, DW>::BYTES;
    static_assert(smem <= 160 * 1024, "LDS budget");
    constexpr int MAX_DEVICES = 64;
    static std::once_flag once[MAX_DEVICES];
    static int cu_count[MAX_DEVICES];
    int dev = 0;
    hip_check(hipGetDevice(&dev), "hipGetDevice");
    if (dev < 0 || dev >= MAX_DEVICES) throw std::runtime_error("dcb_nsplit8: device id out of range");
    std::call_once(once[dev], [&] {
        hipDeviceProp_t prop;
        hip_check(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
            throw std::runtime_error(std::string("dcb_nsplit8 needs gfx950 (160 KB LDS, permlane32_swap); device is ") + prop.gcnArchName);
        }
        hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem),
                  "hipFuncSetAttribute(dcb_nsplit8)");
        int n = 0;
        hip_check(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute");
        cu_count[dev] = n > 0 ? n : 256;
    });
    const int cus = cu_count[dev] * (waves_per_simd<C, CI, PXT, NEXT, DW>() / 2);     // persistent workgroups the chip holds at once
    const int tiles = (p.M + 32 * PXT - 1) / (32 * PXT);
    const int grid = tiles < cus ? tiles : cus;
    hipEvent_t ev0, ev1;
    const int kflop = 6 * CI + (NEXT == 1 ? CI : NEXT);      // 2 M C kflop = the launch's FLOPs (closing conv: + 2 M C NN)
    // (variant: family 5 | inner width | what the NEXT slot holds (0, 1 = dc.0, NN = a closing conv's width) << 12 | depthwise conv inside << 24;
    // the depthwise conv's 18 CI flops per pixel are not counted)
    if (gemm_profile_slot(GemmLaunchInfo{p.M, C, kflop, 0x50000000 | CI | (NEXT << 12) | (DW << 24), 0.f}, &ev0, &ev1)) {
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), smem, stream, ev0, ev1, 0, p);
    } else {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), smem, stream, p);
    }
    hip_check(hipGetLastError(), "dcb_nsplit8 launch");
}

// `next`: 0 = nothing behind ffn.2, 1 = dc.0 of the next block, NN > 1 = the chain's closing conv of width NN (one of FINS...)
template <int C, int CI, int PXT, int... FINS>
void run_px8(const NsParams& p, int next, hipStream_t stream)
{
    if (next == 0) { launch8<C, CI, PXT, 0>(p, stream); return; }
    if (next == 1) { launch8<C, CI, PXT, 1>(p, stream); return; }
    const bool hit = ((next == FINS ? (launch8<C, CI, PXT, FINS>(p, stream), true) : false) || ... || false);
    if (!hit) throw std::invalid_argument("dcb_nsplit8: no instantiation for a closing conv of width " + std::to_string(next));
}

// the same with the block's depthwise conv inside the launch (DW = 1)
template <int C, int CI, int PXT, int... FINS>
void run_px8_dw(const NsParams& p, int next, hipStream_t stream)
{
    if (next == 0) { launch8<C, CI, PXT, 0, 1>(p, stream); return; }
    if (next == 1) { launch8<C, CI, PXT, 1, 1>(p, stream); return; }
    const bool hit = ((next == FINS ? (launch8<C, CI, PXT, FINS, 1>(p, stream), true) : false) || ... || false);
    if (!hit) throw std::invalid_argument("dcb_nsplit8: no instantiation for a closing conv of width " + std::to_string(next));
}

template <int C, int CI, int... FINS>
void run_shape8(const NsParams& p, bool wide, int next, hipStream_t stream)
{
    if constexpr (C < 768) {
        if (wide) { run_px8<C, CI, 2, FINS...>(p, next, stream); return; }
    }
    run_px8<C, CI, 1, FINS...>(p, next, stream);
}

// dcb_nsplit8_<shape>.hip
void run_256_128(const NsParams& p, bool wide, int next, hipStream_t stream);
void run_256_128_dw(const NsParams& p, bool wide, int next, hipStream_t stream);      // depthwise conv inside: dcb_nsplit8_256_128_dw.hip
void run_384_192_dw(const NsParams& p, int next, hipStream_t stream);                 // ... 32-pixel workgroups only: dcb_nsplit8_384_192_dw.hip
void run_256_256(const NsParams& p, bool wide, int next, hipStream_t stream);
void run_192_192(const NsParams& p, bool wide, int next, hipStream_t stream);
void run_384_192(const NsParams& p, bool wide, int next, hipStream_t stream);
void run_384_384(const NsParams& p, bool wide, int next, hipStream_t stream);
void run_512_256(const NsParams& p, bool wide, int next, hipStream_t stream);
void run_512_512(const NsParams& p, bool wide, int next, hipStream_t stream);
void run_768_768(const NsParams& p, bool wide, int next, hipStream_t stream);

}  // namespace nsplit8
}  // namespace dcvc
