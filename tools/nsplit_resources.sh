#!/bin/bash
# registers / scratch of every instantiation of the block kernel (dcb_nsplit8) and of the adaptor + dc.0 kernel (dcb_pair8); no GPU needed
root=$(cd "$(dirname "$0")/.." && pwd)
for f in $root/dcvc_amd/csrc/kernels/dcb_nsplit8_*_*.hip $root/dcvc_amd/csrc/kernels/dcb_pair8_*.hip; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize $DCVC_EXTRA_DEFS -I $root/include -I $root/dcvc_amd/csrc \
      --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs|ScratchSize" | paste - - - - |
    sed -E 's/.*dcb_(nsplit8|pair8)_kernelILi([0-9]+)ELi([0-9]+)ELi([0-9]+)ELi([0-9]+)E(Li([0-9]+)E)?.* VGPRs: ([0-9]+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*/\1 <\2, \3, \4, \5, \7> vgpr=\8 scratch=\9/' ) &
done
wait
