// dcb_pair8_kernel.h (adaptor + dc.0 of a block in one launch) instantiated for the 384-wide blocks
#include "dcb_pair8_kernel.h"

namespace dcvc {
namespace pair8 {

void run_c384(const PairParams& p, int cin, int ci, bool wide, hipStream_t stream)
{
    if (cin == 192 && ci == 384) { run_pair<192, 384, 384>(p, wide, stream); return; }
    throw std::invalid_argument("dcb_pair8: no instantiation for this shape");
}

}  // namespace pair8
}  // namespace dcvc
