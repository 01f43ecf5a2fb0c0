# Same-box A/B/C of whole-library builds on the headline bench: bash tools/ab_libs.sh <out> <rounds> libA.so libB.so ...
# ("default" = the in-tree flair_amd/libflair_hip.so).  Interleaved, <rounds> times.
set -o pipefail
export PYTHONPATH=$PWD
OUT=$1; R=$2; shift 2
mkdir -p $OUT
for r in $(seq $R); do
  for lib in "$@"; do
    if [ $lib = default ]; then unset FLAIR_HIP_LIB; else export FLAIR_HIP_LIB=$PWD/$lib; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>>$OUT/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(l['ms_per_step'],2), 'ms/step')" | tee -a $OUT/bench.log
  done
done
