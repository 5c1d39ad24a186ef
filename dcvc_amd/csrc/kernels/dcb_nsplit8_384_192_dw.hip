// dcb_nsplit8_kernel.h for the (384, 192) blocks - the LD model's prior fusion at picture resolution / 16 - WITH their depthwise
// conv inside the launch (DW = 1; round 6). 32-pixel workgroups only: with 64 pixels LDS has no room for dc.0's output around a tile.
// (a translation unit of its own: the fully unrolled kernels take minutes to compile, the build runs the units in parallel)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

void run_384_192_dw(const NsParams& p, int next, hipStream_t stream)
{
    run_px8_dw<384, 192, 1, 384>(p, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
