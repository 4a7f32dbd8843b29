set -e
O=gpurun_out/r04
mkdir -p $O
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=12 AB_LAUNCHES=20
for v in t4_2_16 walk; do
  python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_$v.so >> $O/ab_walk.log 2>&1
  PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_$v.so >> $O/ab_walk.log 2>&1
done
grep -v amdgpu.ids $O/ab_walk.log
timeout -k 10 100 tools/probes/vol_store_probe.bin 8 | head -14
