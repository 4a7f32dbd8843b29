#!/bin/bash
# A/B builds on full and crop workloads: tools/ab2.sh default libX.so ...
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = default ]; then unset TSDF_HIP_LIB; else export TSDF_ALLOW_LIB_OVERRIDE=1 TSDF_HIP_LIB=$PWD/build/$lib; fi
  for kind in full crop; do
    echo "== $lib $kind (round $rep)"
    PROF_KIND=$kind python3 tools/exp_scale.py 2>&1 | grep "n="
  done
done
done
