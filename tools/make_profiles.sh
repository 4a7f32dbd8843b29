#!/bin/bash
# Run ON THE GPU BOX (via gpurun): profiles of the bench command, written under gpurun_out/profiles_<tag>/.
# Afterwards copy the summaries into profiles/<tag>/ (tracked) with tools/collect_profiles.py.
#   tools/make_profiles.sh r02
# Passes (PMC counters never share a run with anything but --kernel-trace, FETCH_SIZE and WRITE_SIZE
# need separate passes: MI355X_MICROARCH.md, rocprofv3 PMC slots):
#   1. kernel trace + stats of `python3 bench.py --no-extras` -> per-kernel average duration (headline kernel only)
#   2. --pmc FETCH_SIZE   of the same command              -> read-side bytes (x2 for 16-B/lane streams)
#   3. --pmc WRITE_SIZE   of the same command              -> write-side bytes (exact)
#   4. --pmc FETCH_SIZE   of the phase-1-only entry        -> the wide-read share of pass 2
#   5. SQ counters of the same command                     -> where the waves' cycles go
#   6. the augmented 64^3 kernel (BASELINE configs[4]): kernel trace + three SQ passes (tools/gpu_pmc_aug.sh)
#   7. 1024 MSRA-like crops: kernel trace + FETCH/WRITE
set -o pipefail
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
# (a step is 16 launches, each over the next of six resident batches: 6 steps + 1 warm-up + the 48 single launches of the
# spread pass = 160 launches per pass, all of them launches of the rotation; the one-batch pass is skipped)
BENCH="python3 bench.py --steps 6 --warmup 1 --no-extras --no-config3 --no-live-traffic --no-same-batch"
python3 bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err || exit 1
tail -1 $OUT/bench_unprofiled.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
PMC_ROTATE=6 PMC_MODE=aabb rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_aabb -- python3 tools/exp_pmc.py > $OUT/pmc_fetch_aabb.log 2>&1 || { tail -5 $OUT/pmc_fetch_aabb.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- $BENCH --no-cpu-baseline > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; }
python3 tools/collect_profiles.py $OUT summarize
echo "--- augmented 64^3"
tools/gpu_pmc_aug.sh profiles_$TAG/aug64 aug64 > $OUT/aug64.log 2>&1 || tail -3 $OUT/aug64.log
echo "--- plain 64^3"
PMC_TRACE=1 PMC_MODE=r64 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r64_trace -- python3 tools/exp_pmc.py > $OUT/r64_trace.log 2>&1 || tail -3 $OUT/r64_trace.log
echo "--- crops"
PMC_MODE=crop rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/crop_trace -- python3 tools/exp_pmc.py > $OUT/crop_trace.log 2>&1 || tail -3 $OUT/crop_trace.log
PMC_MODE=crop rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/crop_fetch -- python3 tools/exp_pmc.py > $OUT/crop_fetch.log 2>&1 || tail -3 $OUT/crop_fetch.log
PMC_MODE=crop rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/crop_write -- python3 tools/exp_pmc.py > $OUT/crop_write.log 2>&1 || tail -3 $OUT/crop_write.log
python3 tools/collect_profiles.py $OUT summarize2
