#!/bin/bash
# round 5, run 3 (GPU box): straight-line gather address in the augmented pass (no exec-masked branch around the index
# multiply) against HEAD, 64^3-only builds (tools/devbuild.sh dev64_base HEAD / dev64_gidx WORK -DTSDF_DEV_ONLY64), paired.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05; mkdir -p $OUT
{
PROF_R=64 AB_AUG=1 AB_SAME_OUT=1 AB_BLOCKS=16 AB_LAUNCHES=30 python3 tools/ab_precise.py libtsdf_hip_dev64_base.so libtsdf_hip_dev64_gidx.so 2>&1 | grep -v amdgpu.ids
PROF_KIND=crop PROF_R=64 AB_AUG=1 AB_SAME_OUT=1 AB_BLOCKS=12 AB_LAUNCHES=30 python3 tools/ab_precise.py libtsdf_hip_dev64_base.so libtsdf_hip_dev64_gidx.so 2>&1 | grep -v amdgpu.ids
PROF_R=64 AB_SAME_OUT=1 AB_BLOCKS=8 AB_LAUNCHES=30 python3 tools/ab_precise.py libtsdf_hip_dev64_base.so libtsdf_hip_dev64_gidx.so 2>&1 | grep -v amdgpu.ids
} | tee $OUT/ab_gidx.log
