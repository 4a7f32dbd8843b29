#!/bin/bash
# Extra SQ counters for the full fused kernel (GPU box): LDS conflicts, store back-pressure, f64 mix.
set -o pipefail
export TMPDIR=/tmp PMC_MODE=full
OUT=$PWD/gpurun_out/pmc2; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_IFETCH --kernel-trace --output-format csv -d $OUT/a -- python3 tools/exp_pmc.py > $OUT/a.log 2>&1 || tail -3 $OUT/a.log
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/b -- python3 tools/exp_pmc.py > $OUT/b.log 2>&1 || tail -3 $OUT/b.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/c -- python3 tools/exp_pmc.py > $OUT/c.log 2>&1 || tail -3 $OUT/c.log
python3 - <<PY
import csv,glob,collections
for d in ['a','b','c']:
    for f in glob.glob("$OUT/"+d+"/**/*_counter_collection.csv",recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'tsdf_fused_kernel<32, 0, false' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()):
            v=sorted(v); print(k.ljust(34),'median=%.4g'%v[len(v)//2])
PY
