set -e
O=gpurun_out/r04
mkdir -p $O
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=12 AB_LAUNCHES=20
python tools/ab_precise.py libtsdf_hip_dev_now.so libtsdf_hip_dev_hoist.so >> $O/ab_hoist.log 2>&1
PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev_now.so libtsdf_hip_dev_hoist.so >> $O/ab_hoist.log 2>&1
grep -v amdgpu.ids $O/ab_hoist.log
