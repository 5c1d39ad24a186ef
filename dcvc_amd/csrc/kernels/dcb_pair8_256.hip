// dcb_pair8_kernel.h (adaptor + dc.0 of a block in one launch) instantiated for the 256-wide blocks
#include "dcb_pair8_kernel.h"

namespace dcvc {
namespace pair8 {

void run_c256(const PairParams& p, int cin, int ci, bool wide, hipStream_t stream)
{
    if (cin == 448 && ci == 128) { run_pair<448, 256, 128>(p, wide, stream); return; }
    if (cin == 512 && ci == 128) { run_pair<512, 256, 128>(p, wide, stream); return; }
    if (cin == 192 && ci == 128) { run_pair<192, 256, 128>(p, wide, stream); return; }
    if (cin == 128 && ci == 256) { run_pair<128, 256, 256>(p, wide, stream); return; }
    if (cin == 512 && ci == 256) { run_pair<512, 256, 256>(p, wide, stream); return; }
    throw std::invalid_argument("dcb_pair8: no instantiation for this shape");
}

}  // namespace pair8
}  // namespace dcvc
