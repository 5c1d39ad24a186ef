// dcb_nsplit_common.h - what the block kernel (dcb_nsplit8_kernel.h) and the adaptor + dc.0 kernel (dcb_pair8_kernel.h) share:
// the launch parameters, the LDS-DMA piece, the compile-time loop. (Until round 6 these lived in dcb_nsplit_kernel.h, round 3's
// 4-wave form of the block kernel - the 8-wave kernel's A/B partner through rounds 4 and 5, retired in round 6: git history.)
#pragma once
#include "arith.h"
#include "ops.h"
#include "wsilu_table.h"

#include <hip/hip_ext.h>

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace dcvc {

const float4* wsilu_table_device();      // conv_gemm.hip

namespace nsplit {

typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
constexpr int R = 4;                     // interleaved copies of the WSiLU table
constexpr int TABLE_BYTES = WSILU_SEGMENTS * 16;
constexpr int align16k(int bytes) { return (bytes + 16383) & ~16383; }

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N). The weight ring below is indexed ONLY through such
// constants: with plain (later unrolled) loop counters its promotion to registers depended on the order of LLVM's
// unroll / SROA passes and came and went with unrelated edits - 16 x 4 registers through scratch memory, every
// access a vmcnt(0) (measured: the walk at L2 latency).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct NsParams {
    const half_t* t2; int ldt;
    // 8-wave kernel with the depthwise conv inside (DW = 1): dc.0's output [M][ldt] instead of t2, the taps [9][CI], the picture's width
    const half_t* t1; const half_t* wdw; int W;
    const half_t* x; int ldx;
    const half8* wmain;       // packed dc.3 | ffn.0 | ffn.2: per-wave streams of MFMA "A" fragments (dcb_nsplit.hip pack_main8)
    const half8* wnext;       // packed NEXT slot (dc.0 of the next block / the closing conv) or null
    const half_t* b3; const half_t* b0; const half_t* b2; const half_t* b1n;
    const half_t* q; const half_t* q2;
    const half_t* qf;         // 8-wave kernel, closing conv in the NEXT slot (wnext / b1n / t1n / ldt1 are then ITS weights, bias, output): its quant scale or null
    const float4* wsilu;
    half_t* y; int ldy;
    half_t* t1n; int ldt1;
    int M, shortcut;
    long long* timeline;      // optional [workgroups][32] shader-clock stamps of wave 0 (tools/probes/core_bench.hip)
};

// One LDS-DMA piece (64 lanes x 16 B, lane l lands at lds_dst + 16 l), wave-uniform base + 32-bit lane offset.
__device__ __forceinline__ void lds_dma16(const void* sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

}  // namespace nsplit
}  // namespace dcvc
