// dcb_nsplit_kernel.h instantiated for block width 384, inner width 384
#include "dcb_nsplit_kernel.h"

namespace dcvc {
namespace nsplit {

void run_384_384(const NsParams& p, bool wide, bool next, hipStream_t stream)
{
    run_shape<384, 384>(p, wide, next, stream);
}

}  // namespace nsplit
}  // namespace dcvc
