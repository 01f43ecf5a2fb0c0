#!/bin/bash
# HBM traffic of EVERY kernel of a steady-state step: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE: they do not fit one
# pass, MI355X_MICROARCH.md) over a short eager bench run that launches exactly what the steps launch (FLAIR_BENCH_REPLAY=off:
# no isolated replays, whose padding multiplied the serialised launches of a counter pass in round 3).
#   bash tools/pmc_whole_step.sh <outdir> [seconds per pass]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
OUT=$R/${1:-gpurun_out/pmc_whole}
LIMIT=${2:-330}
mkdir -p $OUT
( while sleep 45; do echo "heartbeat $(date +%T)" >> $OUT/heartbeat.log; done ) &
HB=$!
cd /tmp && export TMPDIR=/tmp
export FLAIR_BENCH_REPLAY=off
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pw_$C
  T0=$(date +%s)
  timeout -k 10 $LIMIT rocprofv3 --kernel-trace --pmc $C -d /tmp/pw_$C -o b --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/bench_$C.json 2> $OUT/bench_$C.err
  RC=$?
  T1=$(date +%s)
  N=$(wc -l < /tmp/pw_$C/b_kernel_trace.csv 2>/dev/null || echo 0)
  echo "$C pass: rc=$RC wall=$((T1-T0))s dispatches=$N" | tee -a $OUT/passes.log
  if [ $RC -ne 0 ]; then kill $HB; exit $RC; fi
done
kill $HB
python3 $R/tools/pmc_traffic.py /tmp/pw_FETCH_SIZE /tmp/pw_WRITE_SIZE $OUT/hbm_traffic_pmc_whole_step.json "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph (FLAIR_BENCH_REPLAY=off: every kernel of the eager steps, nothing replayed)" > $OUT/hbm_traffic_pmc_whole_step.txt
head -40 $OUT/hbm_traffic_pmc_whole_step.txt
