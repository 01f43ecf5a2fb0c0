set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r02d
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_blocks_golden.py tests/test_gpu_unet.py tests/test_gpu_sr3.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r02d/tests.log 2>&1; echo "exit=$?" >> gpurun_out/r02d/tests.log; tail -5 gpurun_out/r02d/tests.log
python tools/bench_dcn.py > /dev/null 2>&1
timeout -k 10 400 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r02d/bench30.json 2> gpurun_out/r02d/bench30.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r02d/bench30.json').read().strip().splitlines()[-1])
print(l['value'], l['ms_per_step'], l['roofline']['kernel'], l['roofline']['frac'])
for f in l['roofline']['families']: print(f"{f['family'][:70]:70s} {f['launches']:5d} {f['ms_per_step']:7.2f} ms  frac {f['frac']:.3f}")
PY
