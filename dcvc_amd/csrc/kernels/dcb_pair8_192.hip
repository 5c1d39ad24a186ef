// dcb_pair8_kernel.h (adaptor + dc.0 of a block in one launch) instantiated for the 192-wide block (the intra decoder's last)
#include "dcb_pair8_kernel.h"

namespace dcvc {
namespace pair8 {

void run_c192(const PairParams& p, int cin, int ci, bool wide, hipStream_t stream)
{
    if (cin == 384 && ci == 192) { run_pair<384, 192, 192>(p, wide, stream); return; }
    throw std::invalid_argument("dcb_pair8: no instantiation for this shape");
}

}  // namespace pair8
}  // namespace dcvc
