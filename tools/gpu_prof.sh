#!/bin/bash
# Run on the GPU box via gpurun: kernel trace + PMC passes for tools/prof_phases.py (or bench.py).
# usage: tools/gpu_prof.sh <tag> [script args...]
set -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export PROF_ITERS=5 PROF_KINDS=full
python3 tools/prof_phases.py > $OUT/phases.txt 2>&1 || exit 1
cat $OUT/phases.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_traced.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_phases.py > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/prof_phases.py > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 tools/prof_phases.py > $OUT/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 tools/prof_phases.py > $OUT/pmc_tcc.log 2>&1 || true
find $OUT -name "*.csv" | head -50
