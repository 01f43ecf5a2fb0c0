"""Per-shape read traffic of the dominant kernel from one `rocprofv3 --pmc FETCH_SIZE` pass over tools/pmc_dominant.py (which launches the
shapes of `roofline.by_shape` in order, each `launches` times), beside three models of what the kernel should read:
perfect (every input byte once), halo (every tile re-reads its 18 x 34 halo, temporal taps found in the L2), none (no reuse at all).

    python tools/pmc_by_shape.py <fetch_dir> <bench line json> [raw|x2]
"""
import csv
import json
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1] + "/b_counter_collection.csv"))
        if r["Counter_Name"] == "FETCH_SIZE" and "conv3x3_dma_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
mult = 1.0 if len(sys.argv) > 3 and sys.argv[3] == "raw" else 2.0
print(f"# FETCH_SIZE * 1024 * {mult:g} per launch, MB")
print(f"{'shape':58s} {'n':>4s} {'measured':>9s} {'perfect':>8s} {'halo':>8s} {'none':>8s}")
i = 0
for ent in line["roofline"]["by_shape"]:
    s = ent["shape"]
    n = ent["launches"]
    part = rows[i:i + n]
    i += n
    if not part:
        break
    px = s["T"] * s["H"] * s["W"]
    inb = px * sum(s["cin"]) * 2
    res = px * s["cout"] * 2 * s["residuals"]
    halo = 18 * 34 / (16 * 32)
    tf = (3 * s["T"] - 2) / s["T"] if s["kernel"][0] == 3 else 1
    meas = sum(float(r["Counter_Value"]) for r in part) * 1024 * mult / len(part)
    name = f"T{s['T']} {s['H']}^2 {s['cin']}->{s['cout']} k{s['kernel'][0]} res{s['residuals']}"
    print(f"{name:58s} {n:4d} {meas/1e6:9.1f} {(inb+res)/1e6:8.1f} {(inb*halo+res)/1e6:8.1f} {(inb*halo*tf+res)/1e6:8.1f}")
