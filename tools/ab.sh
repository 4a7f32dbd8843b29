#!/bin/bash
# A/B several builds of the library on the bench workload (GPU box): tools/ab.sh default lib2.so ...
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib (round $rep)"
  if [ "$lib" = default ]; then unset TSDF_HIP_LIB; else export TSDF_ALLOW_LIB_OVERRIDE=1 TSDF_HIP_LIB=$PWD/build/$lib; fi
  python3 tools/exp_scale.py 2>&1 | grep "n="
done
done
