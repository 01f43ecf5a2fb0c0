"""Per-shape kernel durations of tools/bench_conv.py from a rocprofv3 kernel trace."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "conv" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["L0f 224->64", "L0f 64->64", "L0f 64->432", "L0f 192->64", "L1f 416->128", "L1f 128->128", "L1f 128->432",
         "L1f 384->128", "L0c 64->64 2d", "L0c 64->64 3d", "L0c 192->64 2d", "L1c 128 2d", "L1c 128 3d", "L2c 128 3d",
         "L3c 256 3d", "L4c 256 3d", "L5c 512 3d", "L6c 512 3d", "L0c 1x1"]
out = []
for i, n in enumerate(names):
    grp = rows[i * 11:(i + 1) * 11][1:]
    if not grp:
        break
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp]
    out.append(f"{n}:{sum(d)/len(d):.1f}")
print("  ".join(out))
