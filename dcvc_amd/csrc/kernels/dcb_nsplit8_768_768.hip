// dcb_nsplit8_kernel.h instantiated for the (768, 768) blocks (one translation unit per block shape: see dcb_nsplit_kernel.h)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

void run_768_768(const NsParams& p, bool wide, bool next, hipStream_t stream)
{
    run_shape8<768, 768>(p, wide, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
