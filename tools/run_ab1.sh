# bash tools/run_ab1.sh "<pytest -k expr>" ENV=off-value   : parity test, then same-box A/B of one switch
set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/ab1
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "$1" > gpurun_out/ab1/test.log 2>&1 || { tail -30 gpurun_out/ab1/test.log; exit 1; }
tail -2 gpurun_out/ab1/test.log
run() {
  env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>>gpurun_out/ab1/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '->', round(l['ms_per_step'],2), 'ms/step')"
}
run FLAIR_NOOP=1
run $2
run FLAIR_NOOP=1
run $2
