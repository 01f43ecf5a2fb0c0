#!/bin/bash
set -o pipefail
export PYTHONPATH=$PWD
OUT=gpurun_out/r04a; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_face_warp.py -m gpu -x -q -k "dcn or face or warp" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 400 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err && tail -1 $OUT/bench.json | cut -c1-250
timeout -k 10 400 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --attention-resolutions 16,32,64 > $OUT/bench_attn_literal.json 2> $OUT/bench_attn_literal.err && tail -1 $OUT/bench_attn_literal.json | cut -c1-250
bash tools/pmc_whole_step.sh $OUT/pmc 300
