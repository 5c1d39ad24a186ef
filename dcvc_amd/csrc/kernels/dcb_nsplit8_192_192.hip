// dcb_nsplit8_kernel.h instantiated for the (192, 192) block - the intra decoder's last block, dec.dec_2 (round 6: block width
// 192 = six tiles on six of the eight waves, LDS rows of both tensors padded to 256 channels)
#include "dcb_nsplit8_kernel.h"

namespace dcvc {
namespace nsplit8 {

void run_192_192(const NsParams& p, bool wide, int next, hipStream_t stream)
{
    run_shape8<192, 192>(p, wide, next, stream);
}

}  // namespace nsplit8
}  // namespace dcvc
