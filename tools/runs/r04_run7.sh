set -e
O=gpurun_out/r04
mkdir -p $O
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=12 AB_LAUNCHES=20
for v in p4_t4_1_8 ps_p4 ps_p8; do
  python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_$v.so >> $O/ab_phase_shift.log 2>&1
done
PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_ps_p4.so >> $O/ab_phase_shift.log 2>&1
AB_AUG=0 python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_ps_p4.so >> $O/ab_phase_shift.log 2>&1
AB_AUG=0 PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_ps_p4.so >> $O/ab_phase_shift.log 2>&1
grep -v amdgpu.ids $O/ab_phase_shift.log
