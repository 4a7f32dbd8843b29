set -e
O=gpurun_out/r04
mkdir -p $O
export AB_SAME_OUT=1 PROF_R=32 AB_BLOCKS=10 AB_LAUNCHES=40
python tools/ab_precise.py libtsdf_hip_dev32_oldq.so libtsdf_hip_dev32_new3.so >> $O/ab_queue32c.log 2>&1
PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev32_oldq.so libtsdf_hip_dev32_new3.so >> $O/ab_queue32c.log 2>&1
PROF_N=700 PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev32_oldq.so libtsdf_hip_dev32_new3.so >> $O/ab_queue32c.log 2>&1
PROF_R=64 AB_AUG=1 AB_LAUNCHES=20 python tools/ab_precise.py libtsdf_hip_dev_final.so libtsdf_hip_dev_final3.so >> $O/ab_queue32c.log 2>&1
grep -v amdgpu.ids $O/ab_queue32c.log
