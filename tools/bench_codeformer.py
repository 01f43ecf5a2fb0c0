"""CodeFormer auxiliary prior on one window of aligned 512x512 faces: ms per call, conv TFLOP/s
from the per-call HIP events of ops.PROFILE.  (The CPU port is timed by `bench.py --aux codeformer`'s cpu_baseline leg.)

    python tools/bench_codeformer.py [--frames 10] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from flair_amd import ops  # noqa: E402
from flair_amd.guided_diffusion.codeformer import CodeFormer  # noqa: E402
from tests.golden.weights import name_seeded_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10, help="faces per call (the reference's window is 10 frames)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = name_seeded_weights(CodeFormer()).to(dev).eval()
    x = (torch.rand(a.frames, 3, 512, 512, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    res = {"workload": f"CodeFormer(x0, w=1.0, adain=True) on {a.frames} aligned 512x512 faces", "frames": a.frames}
    for name in ("f32", "bf16"):
        if name == "bf16":
            model.convert_to_bf16()
        for _ in range(2):
            model(x, w=1.0, adain=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            model(x, w=1.0, adain=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.iters
        ops.PROFILE = []
        model(x, w=1.0, adain=True)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        fam = {}
        for f, _dt, flops, nbytes, e0, e1, *_rest in prof:
            d = fam.setdefault(f[0], [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += flops
            d[2] += nbytes
            d[3] += e0.elapsed_time(e1)
        conv = fam.get("conv", [0, 0.0, 0.0, 1e-9])
        res[name] = {"ms_per_call": ms, "ms_per_face": ms / a.frames, "conv_launches": conv[0],
                     "conv_gflop_per_face": conv[1] / a.frames / 1e9, "conv_ms_event_sum": conv[3],
                     "conv_tflops": conv[1] / (conv[3] * 1e-3) / 1e12,
                     "whole_call_tflops": conv[1] / (ms * 1e-3) / 1e12,
                     "families_ms": {k: round(v[3], 3) for k, v in fam.items()}}
        print(name, json.dumps(res[name]), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
