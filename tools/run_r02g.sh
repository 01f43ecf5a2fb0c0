set -o pipefail
export PYTHONPATH=$PWD
R=$PWD
mkdir -p gpurun_out/r02g
cd /tmp && export TMPDIR=/tmp
for c in 1 0; do
  rm -rf /tmp/prof$c
  FLAIR_CHAIN=$c FLAIR_DCN_ACT=$c timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$c -o p --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02g/bench_chain$c.json 2>/dev/null
  cp /tmp/prof$c/p_kernel_stats.csv $R/gpurun_out/r02g/kernel_stats_chain$c.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/prof$c/p_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('chain=$c total kernel ms over run:', tot/1e6)
for r in rows[:14]:
    print(f"  {r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
