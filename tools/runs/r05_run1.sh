#!/bin/bash
# round 5, run 1 (GPU box): where the augmented 64^3 kernel's LDS bank conflicts come from.
#   build/libtsdf_hip_dev_r4.so       round 4's kernel ([index][4] inverse-map tables)
#   build/libtsdf_hip_dev_lane.so     lane-major table for the axis the lanes index
#   build/libtsdf_hip_dev_gather0.so  = lane, every depth gather at address 0 (results wrong: the counter is what matters)
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05/lds; mkdir -p $OUT
for v in r4 lane gather0; do
  PROF_AUG=1 PROF_R=64 PROF_K=12 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT \
    --kernel-trace --output-format csv -d $OUT/pmc_$v -- python3 tools/exp_pmc_lib.py libtsdf_hip_dev_$v.so > $OUT/pmc_$v.log 2>&1 || { tail -5 $OUT/pmc_$v.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for v in ("r4", "lane", "gather0"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc_%s/**/*_counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "tsdf_fused_kernel<64, 0, true" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: sorted(x)[len(x) // 2] for k, x in sorted(agg.items())})
PY
PROF_R=64 AB_AUG=1 AB_SAME_OUT=1 AB_BLOCKS=16 AB_LAUNCHES=30 python3 tools/ab_precise.py libtsdf_hip_dev_r4.so libtsdf_hip_dev_lane.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_lane.log
PROF_KIND=crop PROF_R=64 AB_AUG=1 AB_SAME_OUT=1 AB_BLOCKS=12 AB_LAUNCHES=30 python3 tools/ab_precise.py libtsdf_hip_dev_r4.so libtsdf_hip_dev_lane.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_lane.log
