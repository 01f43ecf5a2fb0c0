# SQ counters of the clip-level / per-frame conv kernels on the shapes of config 2 (tools/bench_conv.py):
#   bash tools/pmc_conv.sh <outfile>
set -o pipefail
export PYTHONPATH=$PWD
R=$PWD
OUT=${1:-gpurun_out/conv_pmc_sq.txt}
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_conv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pmc_conv -o d --output-format csv -- python $R/tools/bench_conv.py bf16 clip > /dev/null 2>&1
python - <<'PY' | tee $R/$OUT
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.Counter()
for r in csv.DictReader(open('/tmp/pmc_conv/d_counter_collection.csv')):
    if 'conv' not in r['Kernel_Name']: continue
    k = (r['Kernel_Name'].replace('void (anonymous namespace)::', '')[:60], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('LDS_Block_Size', ''))
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    if r['Counter_Name'] == 'SQ_BUSY_CYCLES': dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("# rocprofv3 --pmc (one pass) -- python tools/bench_conv.py bf16 clip ; values per launch, summed over the chip")
for k in sorted(agg, key=lambda k: -agg[k].get('SQ_WAVE_CYCLES', 0)):
    c = {name: v / n[(k, name)] for name, v in agg[k].items()}
    print(k, "launches", n[(k, 'SQ_WAVE_CYCLES')])
    ns = dur[k] / max(n[(k, 'SQ_BUSY_CYCLES')], 1)
    for name, v in c.items(): print(f"   {name:28s} {v:16.0f}")
    if ns and c.get('SQ_BUSY_CYCLES'):
        print(f"   average launch {ns / 1e3:.1f} us under the profiler; SQ_BUSY_CYCLES / 32 shader engines / duration = {c['SQ_BUSY_CYCLES'] / 32 / ns:.2f} GHz (a lower bound of the clock: busy <= elapsed)")
    if c.get('SQ_BUSY_CYCLES'):
        print(f"   -> MFMA busy / SQ busy = {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / c['SQ_BUSY_CYCLES']:.3f};  LDS-instruction-active / wave cycles = {c.get('SQ_ACTIVE_INST_LDS', 0) / c['SQ_WAVE_CYCLES']:.3f};  waiting on LDS / wave cycles = {c.get('SQ_WAIT_INST_LDS', 0) / c['SQ_WAVE_CYCLES']:.3f}")
PY
