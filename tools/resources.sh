#!/bin/bash
# Register / scratch / LDS use of every kernel instantiation, from the compiler's own remarks:
#   tools/resources.sh [extra -D flags] > profiles/rNN/resources.txt
# (the product's flags; ~3 min: every instantiation is compiled)
cd "$(dirname "$0")/../handposeestimation-with-3d-cnns_amd/csrc" || exit 1
echo "# hipcc -Rpass-analysis=kernel-resource-usage over csrc/tsdf_hip.hip (gfx950, product flags $*)"
echo "# kernel | VGPRs | VGPRs spilled | scratch bytes/lane | occupancy waves/SIMD | LDS bytes"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/tsdf_res.o tsdf_hip.hip 2>&1 |
  awk '/Function Name/ {n=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /VGPRs Spill/ {sp=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {l=$(NF-1); print n, v, sp, s, o, l}' |
  while read -r n v sp s o l; do
    d=$(echo "$n" | c++filt | sed 's/(anonymous namespace):://g; s/^void //; s/(.*//')
    echo "$d | $v | $sp | $s | $o | $l"
  done | sort
