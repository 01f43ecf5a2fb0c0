set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r02f
for v in "1 1 --graph" "0 0 --graph" "1 0 --graph" "1 1 --graph" "1 1"; do
  set -- $v
  FLAIR_CHAIN=$1 FLAIR_DCN_ACT=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 2>gpurun_out/r02f/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chain=$1 dcn_act=$2 $3', round(l['ms_per_step'],2))" | tee -a gpurun_out/r02f/ab.log
done
tail -3 gpurun_out/r02f/err.log
