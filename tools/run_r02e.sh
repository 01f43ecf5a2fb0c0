set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r02e
python tools/bench_dcn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02e/bench_dcn.log
for v in "1 1" "1 0" "0 0" "1 1"; do
  set -- $v
  FLAIR_CHAIN=$1 FLAIR_DCN_ACT=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={x['family'][:28]:round(x['ms_per_step'],2) for x in l['roofline']['families']}
print('chain=$1 dcn_act=$2', round(l['ms_per_step'],2), f)" | tee -a gpurun_out/r02e/ab.log
done
