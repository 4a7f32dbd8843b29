set -e
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 100 tools/probes/vol_store_probe.bin 8 | head -14 | grep -E "^(A|C|D|N)" >> $O/ab_aug_chunks.log 2>&1
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=10 AB_LAUNCHES=20
for v in a2 a1 a4; do
  python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_$v.so >> $O/ab_aug_chunks.log 2>&1
done
python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_base.so >> $O/ab_aug_chunks.log 2>&1
PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_a2.so >> $O/ab_aug_chunks.log 2>&1
grep -v amdgpu.ids $O/ab_aug_chunks.log | tail -20
