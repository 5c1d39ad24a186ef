// dcb_nsplit8_kernel.h instantiated for the (256, 256) blocks (one translation unit per block shape: the fully unrolled kernels take minutes to compile, the build runs the units in parallel)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

// the variants with a chain-closing conv in the NEXT slot: dcb_nsplit8_256_256_fin.hip
extern template void launch8<256, 256, 1, 192>(const NsParams&, hipStream_t);
extern template void launch8<256, 256, 2, 192>(const NsParams&, hipStream_t);

void run_256_256(const NsParams& p, bool wide, int next, hipStream_t stream)
{
    run_shape8<256, 256, 192>(p, wide, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
