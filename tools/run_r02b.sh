set -o pipefail
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r02b
python tools/bench_chain.py > gpurun_out/r02b/bench_chain.log 2>&1; cat gpurun_out/r02b/bench_chain.log
python tools/bench_attn.py --json gpurun_out/r02b/attn_isolated.json > gpurun_out/r02b/bench_attn.log 2>&1; cat gpurun_out/r02b/bench_attn.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/r02b/counters.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM -d /tmp/pmc_dcn -o d --output-format csv -- python $R/tools/bench_dcn.py > /dev/null 2>&1
python - <<'PY' > $R/gpurun_out/r02b/dcn_pmc.txt
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
try:
    for r in csv.DictReader(open('/tmp/pmc_dcn/d_counter_collection.csv')):
        k = r['Kernel_Name'][:60]
        if 'dcn' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k in agg:
        print(k)
        for c, v in agg[k].items(): print(f"   {c:24s} {v / n[(k, c)]:16.0f} per launch")
except Exception as e:
    print('failed', e)
PY
cat $R/gpurun_out/r02b/dcn_pmc.txt
