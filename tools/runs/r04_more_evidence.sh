set -e
O=gpurun_out/r04
mkdir -p $O
export AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=12 AB_LAUNCHES=20
for aug in 1 0; do for kind in full crop; do
  AB_AUG=$aug PROF_KIND=$kind python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_final3.so >> $O/ab_final_vs_static_box2.log 2>&1
done; done
grep -v amdgpu.ids $O/ab_final_vs_static_box2.log
timeout -k 10 900 python tools/fuzz_parity.py 4000 777 > $O/fuzz_parity_4000_rounds_seed777.txt 2>&1; tail -1 $O/fuzz_parity_4000_rounds_seed777.txt
