// dcb_nsplit8_kernel.h for the (384, 192) block that closes the low-delay model's prior fusion chain: y_prior_fusion.conv.3 in the NEXT slot
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

template void launch8<384, 192, 1, 384>(const NsParams&, hipStream_t);
template void launch8<384, 192, 2, 384>(const NsParams&, hipStream_t);

}  // namespace nsplit8
}  // namespace dcvc
