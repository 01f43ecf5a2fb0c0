"""Aggregate a rocprofv3 kernel trace by (kernel, grid, workgroup): calls, mean/total duration."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"]
    if pat not in n:
        continue
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("unsigned short", "bf16")
    short = short.split("(")[0][:60]
    key = (short, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Workgroup_Size_X"]))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[key][0] += 1
    agg[key][1] += d
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k[0]:60s} blocks={k[1]:6d} x{k[2]:4d}  calls={v[0]:5d}  avg={v[1]/v[0]/1e3:8.1f}us  total={v[1]/1e6:8.2f}ms  {100*v[1]/tot:5.1f}%")
print(f"total {tot/1e6:.2f} ms")
