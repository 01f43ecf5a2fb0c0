# Shader clock / power the box reports while the headline bench runs (rocm-smi sampled twice a second beside it): bash tools/clock_under_load.sh <out file>
export PYTHONPATH=$PWD
OUT=${1:-gpurun_out/clock_under_load.txt}
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT.samples &
SMI=$!
timeout -k 10 200 python bench.py --steps 60 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('ms_per_step', l['ms_per_step'])" > $OUT
wait $SMI
cat $OUT.samples >> $OUT
tail -40 $OUT | cut -c1-300
