// dcb_nsplit_kernel.h instantiated for block width 512, inner width 256
#include "dcb_nsplit_kernel.h"

namespace dcvc {
namespace nsplit {

void run_512_256(const NsParams& p, bool wide, bool next, hipStream_t stream)
{
    run_shape<512, 256>(p, wide, next, stream);
}

}  // namespace nsplit
}  // namespace dcvc
