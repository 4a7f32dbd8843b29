set -e
O=gpurun_out/r04
mkdir -p $O
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=8 AB_LAUNCHES=8
for n in 1024 2048 4096; do
  PROF_N=$n python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_dyn4.so >> $O/ab_tail_n.log 2>&1
done
AB_AUG=0 PROF_N=1024 python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_dyn4.so >> $O/ab_tail_n.log 2>&1
AB_AUG=0 PROF_N=4096 python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_dyn4.so >> $O/ab_tail_n.log 2>&1
grep -v amdgpu.ids $O/ab_tail_n.log
