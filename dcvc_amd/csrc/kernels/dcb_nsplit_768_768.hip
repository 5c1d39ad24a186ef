// dcb_nsplit_kernel.h instantiated for block width 768, inner width 768
#include "dcb_nsplit_kernel.h"

namespace dcvc {
namespace nsplit {

void run_768_768(const NsParams& p, bool wide, bool next, hipStream_t stream)
{
    run_shape<768, 768>(p, wide, next, stream);
}

}  // namespace nsplit
}  // namespace dcvc
