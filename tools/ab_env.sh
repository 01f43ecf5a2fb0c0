# Same-box A/B of one environment switch on the headline bench:  bash tools/ab_env.sh <out> <rounds> VAR=value   (against the default)
set -o pipefail
export PYTHONPATH=$PWD
OUT=$1; R=$2; SW=$3
mkdir -p $OUT
for r in $(seq $R); do
  for mode in switch default; do
    if [ $mode = switch ]; then E="$SW"; else E="FLAIR_NOOP=1"; fi
    env $E timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>>$OUT/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$mode ($SW)', round(l['ms_per_step'],2), 'ms/step')" | tee -a $OUT/bench.log
  done
done
