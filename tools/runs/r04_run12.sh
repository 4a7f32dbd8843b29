set -e
O=gpurun_out/r04
mkdir -p $O
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=8 AB_LAUNCHES=20 AB_NOCHECK=1
for v in nogather nodiv noz all; do
  python tools/ab_precise.py libtsdf_hip_dev_k_base.so libtsdf_hip_dev_k_$v.so >> $O/ab_knockouts.log 2>&1
done
AB_AUG=0 python tools/ab_precise.py libtsdf_hip_dev_k_base.so libtsdf_hip_dev_k_base.so >> $O/ab_knockouts.log 2>&1
grep -v amdgpu.ids $O/ab_knockouts.log | tail -20
