#!/bin/bash
# VGPR / AGPR / scratch / occupancy / static LDS / SGPR of every kernel instantiation in libppo_hip.so (compile-only, no GPU):
#   tools/kernel_resources_all.sh > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/.."
echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage, $(hipcc --version | grep -m1 'HIP version')"
echo "# columns: kernel | VGPRs | AGPRs | scratch bytes/lane | occupancy (waves/SIMD, register-limited) | static LDS bytes/block (dynamic LDS is set at launch) | SGPRs"
for f in proximalpolicyoptimization.jl_amd/csrc/*.hip; do
  echo "## $(basename $f)"
  bash tools/kernel_resources.sh $(basename $f) 2>/dev/null | sed 's/^void //'
done
