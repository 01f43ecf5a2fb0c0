#!/bin/bash
# Memory-path counters of the two alignment kernels (tools/bench_dcn.py), separate passes per counter group:  bash tools/pmc_dcn_mem.sh <outfile>
set -o pipefail
export PYTHONPATH=$PWD
R=$PWD
OUT=${1:-gpurun_out/dcn_pmc_mem.txt}
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
: > $R/$OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  # (two more groups of TA_BUFFER_* / TCP_*_STALL_* counters are not collectable on this pool: each attempt sat until its 200 s limit)
  i=$((i+1))
  rm -rf /tmp/pmcd_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcd_$i -o d --output-format csv -- python3 $R/tools/bench_dcn.py > /tmp/pmcd_$i.log 2>&1 || { echo "group $i ($grp) failed: $(tail -2 /tmp/pmcd_$i.log | tr '\n' ' ')" >> $R/$OUT; continue; }
  python3 - $i <<'PY' >> $R/$OUT
import csv, collections, sys
i = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f'/tmp/pmcd_{i}/d_counter_collection.csv')):
    if 'dcn_kernel' not in r['Kernel_Name'] or 'true, true' not in r['Kernel_Name']: continue
    k = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('unsigned short', 'bf16')[:52]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in agg:
    print(k)
    for name, v in agg[k].items(): print(f"   {name:36s} {v / n[(k, name)]:16.0f}   per launch ({n[(k, name)]} launches)")
PY
done
cat $R/$OUT
