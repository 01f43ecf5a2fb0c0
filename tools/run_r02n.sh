set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r02n
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=8 > gpurun_out/r02n/gputests.log 2>&1; echo "exit=$?" >> gpurun_out/r02n/gputests.log; tail -14 gpurun_out/r02n/gputests.log
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>gpurun_out/r02n/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph default:', l['ms_per_step'], l['config']['hip_graph'])"
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-graph 2>>gpurun_out/r02n/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager:', l['ms_per_step'], l['config']['hip_graph'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
