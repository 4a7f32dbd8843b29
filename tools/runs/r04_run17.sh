set -e
O=gpurun_out/r04
mkdir -p $O
export TSDF_ALLOW_LIB_OVERRIDE=1
for rep in 1 2; do
for v in flat xcd; do
  echo "== $v (rep $rep)" >> $O/latency_xcd.log
  TSDF_HIP_LIB=build/libtsdf_hip_dev32_$v.so python tools/latency_table.py 2>&1 | python -c "
import sys, json
t = sys.stdin.read(); j = json.loads(t[t.index('{'):])
print({k: v['back_to_back_us'] for k, v in j.items()})" >> $O/latency_xcd.log
done; done
cat $O/latency_xcd.log
