// dcb_nsplit8_kernel.h instantiated for the (384, 192) blocks - the low-delay model's prior fusion at picture resolution / 16
// (round 6; one translation unit per block shape)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

// the variants with a chain-closing conv in the NEXT slot: dcb_nsplit8_384_192_fin.hip
extern template void launch8<384, 192, 1, 384>(const NsParams&, hipStream_t);
extern template void launch8<384, 192, 2, 384>(const NsParams&, hipStream_t);

void run_384_192(const NsParams& p, bool wide, int next, hipStream_t stream)
{
    run_shape8<384, 192, 384>(p, wide, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
