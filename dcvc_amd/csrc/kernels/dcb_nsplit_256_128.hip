// dcb_nsplit_kernel.h instantiated for block width 256, inner width 128
#include "dcb_nsplit_kernel.h"

namespace dcvc {
namespace nsplit {

void run_256_128(const NsParams& p, bool wide, bool next, hipStream_t stream)
{
    run_shape<256, 128>(p, wide, next, stream);
}

}  // namespace nsplit
}  // namespace dcvc
