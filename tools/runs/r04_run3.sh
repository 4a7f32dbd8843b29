set -e
O=gpurun_out/r04
mkdir -p $O
export AB_AUG=1 AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=16 AB_LAUNCHES=20
for v in pipe4 pipe2 pipe1 pipe8 pipe4c4; do
  python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_$v.so >> $O/ab_pipe.log 2>&1
done
PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_pipe4.so >> $O/ab_pipe.log 2>&1
AB_AUG=0 python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_pipe4.so >> $O/ab_pipe.log 2>&1
AB_AUG=0 PROF_KIND=crop python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_pipe4.so >> $O/ab_pipe.log 2>&1
cat $O/ab_pipe.log | grep -v amdgpu.ids
STAMPS_LIB=build/libtsdf_hip_stamps_pipe4.so python tools/stamps_aug64.py > $O/stamps_aug64_pipe4.log 2>&1
