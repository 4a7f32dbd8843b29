#!/bin/bash
# round 5, run 4 (GPU box): half pools (two LDS half-pool locks; small rectangles stage and voxelize concurrently in both
# groups of a CU) against HEAD, 32^3-only builds (tools/devbuild.sh dev32_base HEAD / dev32_half WORK -DTSDF_DEV_ONLY32).
# Every A/B asserts bit-identical volumes first.  A kernel that hangs is killed by the timeout.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05; mkdir -p $OUT
AB="timeout -k 10 120 python3 tools/ab_precise.py libtsdf_hip_dev32_base.so libtsdf_hip_dev32_half.so"
{
for rot in 6 1; do
  echo "=== AB_ROTATE=$rot"
  AB_ROTATE=$rot PROF_KIND=crop AB_BLOCKS=12 AB_LAUNCHES=36 $AB 2>&1 | grep -v amdgpu.ids || exit 1
  AB_ROTATE=$rot AB_BLOCKS=16 AB_LAUNCHES=36 $AB 2>&1 | grep -v amdgpu.ids || exit 1
done
AB_ROTATE=2 PROF_N=4096 PROF_KIND=crop AB_BLOCKS=8 AB_LAUNCHES=12 $AB 2>&1 | grep -v amdgpu.ids
AB_ROTATE=6 PROF_N=300 AB_BLOCKS=8 AB_LAUNCHES=36 $AB 2>&1 | grep -v amdgpu.ids
AB_ROTATE=6 PROF_N=300 PROF_KIND=crop AB_BLOCKS=8 AB_LAUNCHES=36 $AB 2>&1 | grep -v amdgpu.ids
} | tee $OUT/ab_half_pools.log
