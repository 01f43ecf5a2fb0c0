set -o pipefail
export PYTHONPATH=$PWD
R=$PWD
mkdir -p gpurun_out/r02h
cd /tmp && export TMPDIR=/tmp
for c in 1 0; do
  rm -rf /tmp/prof$c
  FLAIR_CHAIN=$c FLAIR_DCN_ACT=$c timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof$c -o p --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02h/bench_chain$c.json 2>/dev/null
  echo "=== chain=$c" | tee -a $R/gpurun_out/r02h/by_grid.txt
  python $R/tools/trace_by_grid.py /tmp/prof$c/p_kernel_trace.csv "" 30 | tee -a $R/gpurun_out/r02h/by_grid.txt
done
cd $R
for v in "1 1" "0 0" "1 1" "0 0"; do
  set -- $v
  FLAIR_CHAIN=$1 FLAIR_DCN_ACT=$2 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chain=$1 dcn_act=$2', round(l['ms_per_step'],2))" | tee -a gpurun_out/r02h/ab.log
done
