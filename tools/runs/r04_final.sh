set -e
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests_3.log 2>&1 || { tail -30 $O/gpu_tests_3.log; exit 1; }
tail -3 $O/gpu_tests_3.log
python bench.py > $O/bench_2.json 2> $O/bench_2.err
tail -c 400 $O/bench_2.json | head -c 300; echo
export AB_SAME_OUT=1 PROF_R=64 AB_BLOCKS=12 AB_LAUNCHES=20
for aug in 1 0; do for kind in full crop; do
  AB_AUG=$aug PROF_KIND=$kind python tools/ab_precise.py libtsdf_hip_dev_static.so libtsdf_hip_dev_final.so >> $O/ab_final_vs_static.log 2>&1
done; done
grep -v amdgpu.ids $O/ab_final_vs_static.log
python tools/stamps_aug64.py > $O/stamps_aug64_final.log 2>&1
STAMPS_AUG=0 python tools/stamps_aug64.py > $O/stamps_plain64_final.log 2>&1
bash tools/make_profiles.sh r04 > $O/make_profiles.log 2>&1 || tail -20 $O/make_profiles.log
tail -5 $O/make_profiles.log
