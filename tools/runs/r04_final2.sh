set -e
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests_4.log 2>&1 || { tail -30 $O/gpu_tests_4.log; exit 1; }
tail -3 $O/gpu_tests_4.log
python bench.py > $O/bench_3.json 2> $O/bench_3.err
python tools/fuzz_parity.py 400 11 > $O/fuzz_400_seed11.txt 2>&1; tail -1 $O/fuzz_400_seed11.txt
bash tools/make_profiles.sh r04 > $O/make_profiles.log 2>&1 || tail -20 $O/make_profiles.log
tail -3 $O/make_profiles.log
