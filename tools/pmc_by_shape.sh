# bash tools/pmc_by_shape.sh <bench line json> <out txt>: per-shape read traffic of the dominant kernel + calibration of FETCH_SIZE on a plain 1 GiB read
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
export PYTHONPATH=$R
OUT=$R/$2
rm -rf /tmp/pbs /tmp/pcal
timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pbs -o b --output-format csv -- python $R/tools/pmc_dominant.py $R/$1 > /dev/null 2>&1
python $R/tools/pmc_by_shape.py /tmp/pbs $R/$1 > $OUT
timeout -k 10 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pcal -o b --output-format csv -- $R/tools/probes/hbm_peak > /dev/null 2>&1
python - >> $OUT <<'PY'
import csv
print("# calibration: tools/probes/hbm_peak.hip under the same counter (1 GiB buffers, 16 bytes per lane, coalesced): FETCH_SIZE * 1024 / bytes read")
agg = {}
for r in csv.DictReader(open("/tmp/pcal/b_counter_collection.csv")):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    k = r["Kernel_Name"].split("(")[0]
    agg.setdefault(k, []).append(float(r["Counter_Value"]) * 1024)
exp = {"k_read": 1, "k_write": 0, "k_copy": 1, "k_add": 2}
for k, v in agg.items():
    e = exp.get(k.strip(), None)
    print(f"{k:10s} launches {len(v):3d}  FETCH_SIZE*1024 = {sum(v)/len(v)/2**30:.3f} GiB per launch" + (f"  (reads {e} GiB)" if e is not None else ""))
PY
cat $OUT
