// dcb_nsplit8_kernel.h instantiated for the (512, 512) blocks (one translation unit per block shape: the fully unrolled kernels take minutes to compile, the build runs the units in parallel)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

// the variants with a chain-closing conv in the NEXT slot: dcb_nsplit8_512_512_fin.hip
extern template void launch8<512, 512, 1, 256>(const NsParams&, hipStream_t);
extern template void launch8<512, 512, 2, 256>(const NsParams&, hipStream_t);
extern template void launch8<512, 512, 1, 512>(const NsParams&, hipStream_t);
extern template void launch8<512, 512, 2, 512>(const NsParams&, hipStream_t);

void run_512_512(const NsParams& p, bool wide, int next, hipStream_t stream)
{
    run_shape8<512, 512, 256, 512>(p, wide, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
