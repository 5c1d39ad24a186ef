// dcb_nsplit8_kernel.h for the (256, 128) blocks with their depthwise conv inside that close a chain
// (a translation unit of its own: the fully unrolled kernels take minutes to compile, the build runs the units in parallel)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

template void launch8<256, 128, 1, 128, 1>(const NsParams&, hipStream_t);
template void launch8<256, 128, 2, 128, 1>(const NsParams&, hipStream_t);
template void launch8<256, 128, 1, 192, 1>(const NsParams&, hipStream_t);
template void launch8<256, 128, 2, 192, 1>(const NsParams&, hipStream_t);
template void launch8<256, 128, 1, 256, 1>(const NsParams&, hipStream_t);
template void launch8<256, 128, 2, 256, 1>(const NsParams&, hipStream_t);

}  // namespace nsplit8
}  // namespace dcvc
