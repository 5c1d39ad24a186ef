#!/bin/bash
# registers / scratch of every instantiation of the N-split block kernel (no GPU needed)
root=$(cd "$(dirname "$0")/.." && pwd)
for f in $root/dcvc_amd/csrc/kernels/dcb_nsplit_*_*.hip; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize $DCVC_EXTRA_DEFS -I $root/include -I $root/dcvc_amd/csrc \
      --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs|ScratchSize" | paste - - - - |
    sed -E 's/.*dcb_nsplit_kernelILi([0-9]+)ELi([0-9]+)ELi([0-9]+)ELb([01]).* VGPRs: ([0-9]+).*AGPRs: ([0-9]+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*/C=\1 CI=\2 PXT=\3 NEXT=\4 vgpr=\5 agpr=\6 scratch=\7/' ) &
done
wait
