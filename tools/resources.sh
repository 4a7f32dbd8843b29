#!/bin/bash
# Register / scratch / LDS use of every kernel instantiation (hipcc remarks): tools/resources.sh [extra -D flags]
cd "$(dirname "$0")/../handposeestimation-with-3d-cnns_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/tsdf_res.o tsdf_hip.hip 2>&1 |
  awk '/Function Name/ {n=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /VGPRs Spill/ {sp=$(NF-1)} /LDS Size/ {l=$(NF-1); print n, "VGPRs", v, "spilled", sp, "scratch", s, "LDS", l}'
