# rocprofv3 kernel trace of the CodeFormer prior bench, grouped by (kernel, grid):  bash tools/trace_codeformer.sh <outfile>
set -o pipefail
R=$PWD
export PYTHONPATH=$R
OUT=$R/${1:-gpurun_out/codeformer_by_grid.txt}
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cfp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/cfp -o p --output-format csv -- python $R/tools/bench_codeformer.py --iters 3 > /dev/null 2>&1
python $R/tools/trace_by_grid.py /tmp/cfp/p_kernel_trace.csv "" 60 > $OUT
head -40 $OUT
