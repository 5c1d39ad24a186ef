// Stand-in for dcb_core.hip (round 2's block kernel: activations in registers, weight slabs through an LDS ring) in the
// default build. Since round 4 the kernel is two generations old - dcb_nsplit8 runs every block shape, dcb_nsplit is its
// A/B partner - and is compiled only on request:  DCVC_EXTRA_DEFS=-DDCVC_WITH_DCB_CORE python -m dcvc_amd.build
// (dcvc_amd/build.py picks dcb_core.hip instead of this file). The entry points stay in the ABI and say so.
#include "ops.h"

#include <stdexcept>

namespace dcvc {

bool dcb_core_supported(int, int, int) { return false; }

void dcb_core_timeline_buffer(long long*) {}

void dcb_core(const DcbCoreDesc&, hipStream_t)
{
    throw std::runtime_error("dcb_core is not part of this build (DCVC_EXTRA_DEFS=-DDCVC_WITH_DCB_CORE python -m dcvc_amd.build)");
}

}  // namespace dcvc
