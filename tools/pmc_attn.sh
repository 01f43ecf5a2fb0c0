set -o pipefail
export PYTHONPATH=$PWD
R=$PWD
mkdir -p gpurun_out/r02k
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d /tmp/pmc_attn -o d --output-format csv -- python $R/tools/bench_attn.py > /dev/null 2>&1
python - <<'PY' | tee $R/gpurun_out/r02k/attn_pmc.txt
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open('/tmp/pmc_attn/d_counter_collection.csv')):
    if 'attn' not in r['Kernel_Name']: continue
    k = (r['Kernel_Name'][:70], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', ''))
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in agg:
    print(k)
    for c, v in agg[k].items(): print(f"   {c:28s} {v / n[(k, c)]:16.0f} per launch")
PY
