# Kernel trace of a short bench run, aggregated by (kernel, grid): bash tools/run_trace_all.sh <outdir>
set -o pipefail
R=$PWD
export PYTHONPATH=$R
OUT=$R/${1:-gpurun_out/trace_all}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tall
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tall -o p --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graph > $OUT/bench.json 2>/dev/null
python $R/tools/trace_by_grid.py /tmp/tall/p_kernel_trace.csv "" 90 > $OUT/by_grid.txt
tail -1 $OUT/bench.json | cut -c1-200
