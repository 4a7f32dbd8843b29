#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> <mode aabb|full> -- SQ / TCC counters for the fused kernel
set -o pipefail
TAG=${1:-pmc}; export PMC_MODE=${2:-aabb}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/sq -- python3 tools/exp_pmc.py > $OUT/sq.log 2>&1 || { tail -3 $OUT/sq.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq2 -- python3 tools/exp_pmc.py > $OUT/sq2.log 2>&1 || { tail -3 $OUT/sq2.log; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 tools/exp_pmc.py > $OUT/fetch.log 2>&1 || { tail -3 $OUT/fetch.log; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 tools/exp_pmc.py > $OUT/write.log 2>&1 || { tail -3 $OUT/write.log; }
python3 - <<PY
import csv,glob,collections
for d in ['sq','sq2','fetch','write']:
    fs=glob.glob("$OUT/"+d+"/**/*_counter_collection.csv",recursive=True)
    if not fs: continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'tsdf' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[1][-22:] if False else r['Kernel_Name'][27:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()):
        v=sorted(v); print(d,k[0],k[1],'n=%d'%len(v),'median=%.4g'%v[len(v)//2])
PY
