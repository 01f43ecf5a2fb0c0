#!/bin/bash
# Lower bound of the shader clock inside every kernel of the steady-state steps: one rocprofv3 pass with SQ_BUSY_CYCLES (+ the matrix pipe's
# busy cycles) over the eager bench run of tools/pmc_whole_step.sh; clock >= SQ_BUSY_CYCLES / 32 shader engines / launch duration.
#   bash tools/pmc_clock_whole_step.sh <out txt>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
OUT=$R/${1:-gpurun_out/clock_whole_step.txt}
cd /tmp && export TMPDIR=/tmp
export FLAIR_BENCH_REPLAY=off
rm -rf /tmp/pw_clk
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/pw_clk -o b --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > /dev/null 2>&1
python3 - > $OUT <<'PY'
import collections, csv, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.Counter()
for r in csv.DictReader(open('/tmp/pw_clk/b_counter_collection.csv')):
    m = re.search(r"namespace\)::(\w+(?:<[^>]*>)?)", r["Kernel_Name"])
    k = ((m.group(1) if m else r["Kernel_Name"][:50]).replace("unsigned short", "bf16"), r.get("Grid_Size", ""))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        n[k] += 1
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("# rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph (FLAIR_BENCH_REPLAY=off)")
print("# clock >= SQ_BUSY_CYCLES / 32 shader engines / duration;  matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (SQ_BUSY_CYCLES / 32)")
print(f"{'kernel':58s} {'grid':>9s} {'calls':>6s} {'avg us':>8s} {'>= GHz':>7s} {'MFMA busy':>10s}")
for k in sorted(agg, key=lambda k: -dur[k])[:28]:
    c = agg[k]
    us = dur[k] / n[k] / 1e3
    ghz = c["SQ_BUSY_CYCLES"] / n[k] / 32 / (dur[k] / n[k])
    mf = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / max(c["SQ_BUSY_CYCLES"] / 32, 1)
    print(f"{k[0]:58s} {k[1]:>9s} {n[k]:6d} {us:8.1f} {ghz:7.2f} {mf:10.2f}")
PY
cat $OUT
