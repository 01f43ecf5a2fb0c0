set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/gnab
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "group_norm" > gpurun_out/gnab/test.log 2>&1 || { tail -30 gpurun_out/gnab/test.log; exit 1; }
tail -2 gpurun_out/gnab/test.log
echo fused; timeout -k 10 200 python tools/bench_gn.py
echo three-launch; FLAIR_GN_FUSED_MAX=0 timeout -k 10 200 python tools/bench_gn.py
run() {
  env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>>gpurun_out/gnab/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '->', round(l['ms_per_step'],2), 'ms/step')"
}
run FLAIR_NOOP=1
run FLAIR_GN_FUSED_MAX=0
run FLAIR_NOOP=1
run FLAIR_GN_FUSED_MAX=0
