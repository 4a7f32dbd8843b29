#!/bin/bash
# SQ counters + kernel trace of the augmented kernel (GPU box).  usage: tools/gpu_pmc_aug.sh <outdir-tag> [aug64|aug32]
# Counters go in their own passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -o pipefail
TAG=${1:-pmc_aug}; MODE=${2:-aug64}
export TMPDIR=/tmp PMC_MODE=$MODE
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
PMC_TRACE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/exp_pmc.py > $OUT/trace.log 2>&1 || { tail -3 $OUT/trace.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/sq_a -- python3 tools/exp_pmc.py > $OUT/a.log 2>&1 || tail -3 $OUT/a.log
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/sq_b -- python3 tools/exp_pmc.py > $OUT/b.log 2>&1 || tail -3 $OUT/b.log
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_WR --kernel-trace --output-format csv -d $OUT/sq_c -- python3 tools/exp_pmc.py > $OUT/c.log 2>&1 || tail -3 $OUT/c.log
python3 - <<PY
import csv,glob,collections,json
res={}
for f in glob.glob("$OUT/trace/**/*_kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tsdf' in r['Name']:
            res.setdefault('kernel_trace',{})[r['Name']]={'calls':int(r['Calls']),'avg_us':float(r['AverageNs'])/1e3,'min_us':float(r['MinNs'])/1e3}
# steady state: the average over dispatches 41.. of each kernel (see tools/exp_pmc.py)
for f in glob.glob("$OUT/trace/**/*_kernel_trace.csv",recursive=True):
    per=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tsdf' in r['Kernel_Name']:
            per[r['Kernel_Name']].append((int(r['Start_Timestamp']),int(r['End_Timestamp'])))
    for kn,v in per.items():
        v.sort()
        d=[(e-s)/1e3 for s,e in v]
        if len(d)>60 and kn in res.get('kernel_trace',{}):
            t=d[40:]
            res['kernel_trace'][kn].update({'steady_calls':len(t),'steady_avg_us':sum(t)/len(t),'first_10_avg_us':sum(d[:10])/10})
for d in ['sq_a','sq_b','sq_c']:
    for f in glob.glob("$OUT/"+d+"/**/*_counter_collection.csv",recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'tsdf' in r['Kernel_Name']:
                agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
        for kn,cs in agg.items():
            for k,v in sorted(cs.items()):
                v=sorted(v); res.setdefault('counters',{}).setdefault(kn,{})[k]=v[len(v)//2]
json.dump(res,open("$OUT/summary.json","w"),indent=1)
print(json.dumps(res,indent=1))
PY
