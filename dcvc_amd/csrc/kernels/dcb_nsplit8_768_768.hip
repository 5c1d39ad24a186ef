// dcb_nsplit8_kernel.h instantiated for the (768, 768) blocks (one translation unit per block shape: the fully unrolled kernels take minutes to compile, the build runs the units in parallel)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

// the variants with a chain-closing conv in the NEXT slot: dcb_nsplit8_768_768_fin.hip
extern template void launch8<768, 768, 1, 768>(const NsParams&, hipStream_t);

void run_768_768(const NsParams& p, bool wide, int next, hipStream_t stream)
{
    run_shape8<768, 768, 768>(p, wide, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
