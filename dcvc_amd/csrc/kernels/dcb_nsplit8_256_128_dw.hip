// dcb_nsplit8_kernel.h for the (256, 128) blocks WITH their depthwise conv inside the launch (DW = 1; round 6): the inter models'
// blocks at picture resolution / 8 and / 16 - a launch and a round trip of the 128-channel tensor through memory less per block
// (a translation unit of its own: the fully unrolled kernels take minutes to compile, the build runs the units in parallel)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

// the variants with a chain-closing conv in the NEXT slot: dcb_nsplit8_256_128_dw_fin.hip
extern template void launch8<256, 128, 1, 128, 1>(const NsParams&, hipStream_t);
extern template void launch8<256, 128, 2, 128, 1>(const NsParams&, hipStream_t);
extern template void launch8<256, 128, 1, 192, 1>(const NsParams&, hipStream_t);
extern template void launch8<256, 128, 2, 192, 1>(const NsParams&, hipStream_t);
extern template void launch8<256, 128, 1, 256, 1>(const NsParams&, hipStream_t);
extern template void launch8<256, 128, 2, 256, 1>(const NsParams&, hipStream_t);

void run_256_128_dw(const NsParams& p, bool wide, int next, hipStream_t stream)
{
    if (wide) run_px8_dw<256, 128, 2, 128, 192, 256>(p, next, stream);
    else run_px8_dw<256, 128, 1, 128, 192, 256>(p, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
