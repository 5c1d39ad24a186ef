// dcb_pair8_kernel.h (adaptor + dc.0 of a block in one launch) instantiated for the 512-wide blocks
#include "dcb_pair8_kernel.h"

namespace dcvc {
namespace pair8 {

void run_c512(const PairParams& p, int cin, int ci, bool wide, hipStream_t stream)
{
    if (cin == 256 && ci == 512) { run_pair<256, 512, 512>(p, wide, stream); return; }
    if (cin == 512 && ci == 512) { run_pair<512, 512, 512>(p, wide, stream); return; }
    if (cin == 192 && ci == 512) { run_pair<192, 512, 512>(p, wide, stream); return; }
    if (cin == 192 && ci == 256) { run_pair<192, 512, 256>(p, wide, stream); return; }
    throw std::invalid_argument("dcb_pair8: no instantiation for this shape");
}

}  // namespace pair8
}  // namespace dcvc
