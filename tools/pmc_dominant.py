"""Launches every shape of the dominant conv kernel exactly as often as one denoising step does (shapes and counts from the
`roofline.by_shape` list of a bench line), eagerly and nothing else: under `rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE` the
per-kernel average of the process is then the launch-count-weighted HBM traffic per launch (tools/pmc_traffic.py).

    python tools/pmc_dominant.py profiles/r03z_bench_full.json
"""
import json
import sys

import torch

from flair_amd import ops

line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
dev = torch.device("cuda:0")
dt = torch.bfloat16
ACT = {0: ops.ACT_NONE, 1: ops.ACT_RELU, 2: ops.ACT_LRELU01, 3: ops.ACT_SILU}
n = 0
for ent in line["roofline"]["by_shape"]:
    s = ent["shape"]
    T, H, W, cins, cout, k = s["T"], s["H"], s["W"], s["cin"], s["cout"], tuple(s["kernel"])
    xs = [torch.randn(T, H, W, c, device=dev).to(dt) for c in cins]
    taps, cin = k[0] * k[1] * k[2], sum(cins)
    w = (torch.randn(cout, taps, cin, device=dev) / (taps * cin) ** 0.5).to(dt)
    b = torch.randn(cout, device=dev)
    kw = {}
    if s["residuals"] >= 1:
        kw["res0"] = torch.randn(T, H, W, cout, device=dev).to(dt)
    if s["residuals"] >= 2:
        kw["res1"] = torch.randn(T, H, W, cout, device=dev).to(dt)
    act = s["act"]
    if act == 4:                                   # FLAIR_ACT_DCN_OFFSETS (c -> 27 * G offsets / masks)
        kw.update(act=4, act_param=10.0, act_period=48)
    else:
        kw["act"] = ACT.get(act, ops.ACT_NONE)
    y = ops.conv(xs, w, b, cout, k, **kw)
    for _ in range(ent["launches"] - 1):
        ops.conv(xs, w, b, cout, k, out=y, **kw)
    n += ent["launches"]
torch.cuda.synchronize()
print(f"{n} launches of {len(line['roofline']['by_shape'])} shapes")
