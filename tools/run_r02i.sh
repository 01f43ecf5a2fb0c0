set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r02i
timeout -k 10 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_sr3.py tests/test_blocks_golden.py tests/test_video_windows.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r02i/tests.log 2>&1; echo "exit=$?" >> gpurun_out/r02i/tests.log; tail -4 gpurun_out/r02i/tests.log
for g in "" "--graph"; do
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline $g 2>gpurun_out/r02i/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
f={x['family'][:24]:round(x['ms_per_step'],2) for x in l['roofline']['families']}
print('$g', round(l['ms_per_step'],2), f)" | tee -a gpurun_out/r02i/bench.log
done
