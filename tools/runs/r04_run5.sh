set -e
O=gpurun_out/r04
mkdir -p $O
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=12 AB_LAUNCHES=20
for v in dyn4 dyn4_st20 dyn4_st45 dyn2 static_st45; do
  python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_$v.so >> $O/ab_stagger64.log 2>&1
done
PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_dyn4.so >> $O/ab_stagger64.log 2>&1
AB_AUG=0 python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_static_st45.so >> $O/ab_stagger64.log 2>&1
grep -v amdgpu.ids $O/ab_stagger64.log
