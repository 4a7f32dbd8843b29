#!/bin/bash
# Per-kernel durations of the bench workload (rocprofv3 kernel trace).  usage: tools/gpu_trace.sh <tag>
set -o pipefail
TAG=${1:-trace}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_traced.log 2>&1 || { tail -5 $OUT/bench_traced.log; exit 1; }
tail -1 $OUT/bench_traced.log | cut -c1-160
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'tsdf' in r['Name']:
        print(r['Name'][:70].ljust(72), 'calls',r['Calls'],'avg_us %.1f'%(float(r['AverageNs'])/1e3),'min_us %.1f'%(float(r['MinNs'])/1e3),'max_us %.1f'%(float(r['MaxNs'])/1e3))
PY
