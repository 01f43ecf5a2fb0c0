set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r02c
timeout -k 10 400 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r02c/bench30.json 2> gpurun_out/r02c/bench30.err; tail -c 3000 gpurun_out/r02c/bench30.json
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r02c/gputests.log 2>&1; echo "exit=$?" >> gpurun_out/r02c/gputests.log; tail -5 gpurun_out/r02c/gputests.log
