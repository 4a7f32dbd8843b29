set -e
mkdir -p gpurun_out/r04
O=gpurun_out/r04
STAMPS_LIB=build/libtsdf_hip_stamps_static.so python tools/stamps_aug64.py > $O/stamps_aug64_static_waves.log 2>&1
STAMPS_LIB=build/libtsdf_hip_stamps_dyn8.so python tools/stamps_aug64.py > $O/stamps_aug64_dyn8_waves.log 2>&1
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=16 AB_LAUNCHES=20
for v in dyn8 dyn16 dyn4; do
  python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_$v.so >> $O/ab_dyn_tiles.log 2>&1
done
PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_dyn8.so >> $O/ab_dyn_tiles.log 2>&1
cat $O/ab_dyn_tiles.log
