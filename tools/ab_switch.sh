# bash tools/ab_switch.sh "<pytest -k expr>" "ENV=on-value"   : parity test WITH the switch, then same-box A/B (default vs switch)
set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/ab_switch
env $2 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "$1" > gpurun_out/ab_switch/test.log 2>&1 || { tail -30 gpurun_out/ab_switch/test.log; exit 1; }
tail -2 gpurun_out/ab_switch/test.log
run() {
  env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>>gpurun_out/ab_switch/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '->', round(l['ms_per_step'],2), 'ms/step')"
}
run FLAIR_NOOP=1
run $2
run FLAIR_NOOP=1
run $2
