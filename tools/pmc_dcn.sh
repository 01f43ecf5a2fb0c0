#!/bin/bash
# SQ counters of the two alignment kernels on the per-frame shapes of config 2 (tools/bench_dcn.py):  bash tools/pmc_dcn.sh <outfile>
set -o pipefail
export PYTHONPATH=$PWD
R=$PWD
OUT=${1:-gpurun_out/dcn_pmc_sq.txt}
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_dcn
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/pmc_dcn -o d --output-format csv -- python3 $R/tools/bench_dcn.py > /dev/null 2>&1
python3 - <<'PY' | tee $R/$OUT
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open('/tmp/pmc_dcn/d_counter_collection.csv')):
    if 'dcn_kernel' not in r['Kernel_Name']: continue
    k = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('unsigned short', 'bf16')[:70]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
print("# rocprofv3 --pmc (one pass, 8 SQ counters) -- python tools/bench_dcn.py ; values per launch, summed over the chip")
for k in sorted(agg, key=lambda k: -agg[k].get('SQ_WAVE_CYCLES', 0)):
    c = {name: v / n[(k, name)] for name, v in agg[k].items()}
    print(k, "launches", n[(k, 'SQ_WAVE_CYCLES')])
    for name, v in c.items(): print(f"   {name:24s} {v:16.0f}")
    if c.get('SQ_WAVE_CYCLES'):
        print(f"   -> of the wave cycles: waiting {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}, issue-stalled {c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}, "
              f"issuing {c.get('SQ_ACTIVE_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f} (VALU {c.get('SQ_ACTIVE_INST_VALU', 0) / c['SQ_WAVE_CYCLES']:.3f}); "
              f"VALU share of issue cycles {c.get('SQ_ACTIVE_INST_VALU', 0) / max(c.get('SQ_ACTIVE_INST_ANY', 1), 1):.3f}")
PY
