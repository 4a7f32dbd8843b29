#!/bin/bash
# SQ counters of two builds of the library on the same workload (GPU box): tools/gpu_pmc_ab.sh <tag> libA.so libB.so
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
for lib in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH --kernel-trace --output-format csv -d $OUT/a$i -- python3 tools/exp_pmc_lib.py $lib > $OUT/a$i.log 2>&1 || tail -3 $OUT/a$i.log
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/b$i -- python3 tools/exp_pmc_lib.py $lib > $OUT/b$i.log 2>&1 || tail -3 $OUT/b$i.log
  rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_WAVE32_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/c$i -- python3 tools/exp_pmc_lib.py $lib > $OUT/c$i.log 2>&1 || tail -3 $OUT/c$i.log
done
python3 - "$OUT" "$@" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]; libs=sys.argv[2:]
tab=collections.defaultdict(dict)
for i,lib in enumerate(libs,1):
    for d in 'abc':
        for f in glob.glob(f"{out}/{d}{i}/**/*_counter_collection.csv",recursive=True):
            agg=collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'tsdf_fused' in r['Kernel_Name']:
                    agg[r['Counter_Name']].append(float(r['Counter_Value']))
            for k,v in agg.items():
                v=sorted(v); tab[k][lib]=v[len(v)//2]
print("counter".ljust(28), *[l[-22:].rjust(24) for l in libs])
for k in sorted(tab):
    print(k.ljust(28), *[("%.4g"%tab[k].get(l,float('nan'))).rjust(24) for l in libs])
PY
