// dcb_nsplit8_kernel.h for the (768, 768) blocks that close a chain: the chain's last 1x1 conv in the NEXT slot
// (a translation unit of its own: the fully unrolled kernels take minutes to compile, the build runs the units in parallel)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

template void launch8<768, 768, 1, 768>(const NsParams&, hipStream_t);

}  // namespace nsplit8
}  // namespace dcvc
