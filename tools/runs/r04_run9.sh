set -e
O=gpurun_out/r04
mkdir -p $O
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=10 AB_LAUNCHES=20 AB_NOCHECK=1
for v in noxy nop1 noxy_nop1; do
  python tools/ab_precise.py libtsdf_hip_dev_base.so libtsdf_hip_dev_$v.so >> $O/ab_timing_only.log 2>&1
done
grep -v amdgpu.ids $O/ab_timing_only.log
