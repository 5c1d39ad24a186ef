// dcb_nsplit8_kernel.h instantiated for the (512, 256) blocks (one translation unit per block shape: the fully unrolled kernels take minutes to compile, the build runs the units in parallel)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

void run_512_256(const NsParams& p, bool wide, int next, hipStream_t stream)
{
    run_shape8<512, 256>(p, wide, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
