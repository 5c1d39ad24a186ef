// dcb_nsplit8_kernel.h instantiated for the (512, 512) blocks (one translation unit per block shape: see dcb_nsplit_kernel.h)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

void run_512_512(const NsParams& p, bool wide, bool next, hipStream_t stream)
{
    run_shape8<512, 512>(p, wide, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
