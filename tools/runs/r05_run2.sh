#!/bin/bash
# round 5, run 2 (GPU box): head help (both groups of a CU stream its first frame) against the same tree without it,
# 32^3-only builds (tools/devbuild.sh dev32_base HEAD / dev32_head WORK -DTSDF_DEV_ONLY32), paired, bit-identical.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05; mkdir -p $OUT
{
for rot in 6 1; do
  echo "=== AB_ROTATE=$rot"
  AB_ROTATE=$rot AB_BLOCKS=16 AB_LAUNCHES=36 python3 tools/ab_precise.py libtsdf_hip_dev32_base.so libtsdf_hip_dev32_head.so 2>&1 | grep -v amdgpu.ids
  AB_ROTATE=$rot PROF_KIND=crop AB_BLOCKS=12 AB_LAUNCHES=36 python3 tools/ab_precise.py libtsdf_hip_dev32_base.so libtsdf_hip_dev32_head.so 2>&1 | grep -v amdgpu.ids
done
AB_ROTATE=2 PROF_N=4096 AB_BLOCKS=8 AB_LAUNCHES=12 python3 tools/ab_precise.py libtsdf_hip_dev32_base.so libtsdf_hip_dev32_head.so 2>&1 | grep -v amdgpu.ids
AB_ROTATE=6 PROF_N=300 AB_BLOCKS=8 AB_LAUNCHES=36 python3 tools/ab_precise.py libtsdf_hip_dev32_base.so libtsdf_hip_dev32_head.so 2>&1 | grep -v amdgpu.ids
} | tee $OUT/ab_head.log
