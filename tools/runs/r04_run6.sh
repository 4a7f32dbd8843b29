set -e
O=gpurun_out/r04
mkdir -p $O
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=12 AB_LAUNCHES=20
for v in dyn4 t8_2_16 t4_1_8 t4_2_16 p4_t4_1_8 p8_t4_1_8; do
  python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_$v.so >> $O/ab_tiers.log 2>&1
done
grep -v amdgpu.ids $O/ab_tiers.log
STAMPS_LIB=build/libtsdf_hip_stamps_p4_t4_1_8.so python tools/stamps_aug64.py > $O/stamps_aug64_p4_t4_1_8.log 2>&1
