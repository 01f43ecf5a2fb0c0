# Memory-path counters of the 1x1 128 -> 64 convolution on 16 x 256^2 (conv_igemm_kernel<64,128>) beside gn_apply on the same tensor
# size:  bash tools/pmc_1x1.sh <outfile>      (separate passes per counter group: the TCC / TA groups do not share a pass)
set -o pipefail
export PYTHONPATH=$PWD
R=$PWD
OUT=${1:-gpurun_out/pmc_1x1.txt}
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
: > $R/$OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_WRREQ_STALL_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc1_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc1_$i -o d --output-format csv -- python $R/tools/bench_conv.py bf16 "1x1" > /tmp/pmc1_$i.log 2>&1 || { echo "group $i failed: $(tail -2 /tmp/pmc1_$i.log)" >> $R/$OUT; continue; }
  python - $i <<'PY' >> $R/$OUT
import csv, collections, sys
i = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f'/tmp/pmc1_{i}/d_counter_collection.csv')):
    if 'igemm' not in r['Kernel_Name']: continue
    k = r['Kernel_Name'].replace('void (anonymous namespace)::', '')[:50]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in agg:
    print(k)
    for name, v in agg[k].items(): print(f"   {name:32s} {v / n[(k, name)]:16.0f}   per launch ({n[(k, name)]} launches)")
PY
done
cat $R/$OUT
