"""Micro-benchmark of the conv kernel on the shapes of one UNet forward (config 2)."""
import os
import sys
import torch
from flair_amd import ops

SHAPES = [
    # name, T, H, W, segs, cout, kernel
    ("L0 frame 224->64", 1, 256, 256, [64, 64, 64, 32], 64, (1, 3, 3)),
    ("L0 frame 64->64", 1, 256, 256, [64], 64, (1, 3, 3)),
    ("L0 frame 64->432", 1, 256, 256, [64], 432, (1, 3, 3)),
    ("L0 frame 192->64", 1, 256, 256, [64, 64, 64], 64, (1, 3, 3)),
    ("L1 frame 416->128", 1, 128, 128, [128, 128, 128, 32], 128, (1, 3, 3)),
    ("L1 frame 128->128", 1, 128, 128, [128], 128, (1, 3, 3)),
    ("L1 frame 128->432", 1, 128, 128, [128], 432, (1, 3, 3)),
    ("L1 frame 384->128", 1, 128, 128, [128, 128, 128], 128, (1, 3, 3)),
    ("L0 clip 64->64 2d", 16, 256, 256, [64], 64, (1, 3, 3)),
    ("L0 clip 64->64 3d", 16, 256, 256, [64], 64, (3, 3, 3)),
    ("L0 clip 192->64 2d", 16, 256, 256, [64, 64, 64], 64, (1, 3, 3)),
    ("L1 clip 128->128 2d", 16, 128, 128, [128], 128, (1, 3, 3)),
    ("L1 clip 128->128 3d", 16, 128, 128, [128], 128, (3, 3, 3)),
    ("L2 clip 128->128 3d", 16, 64, 64, [128], 128, (3, 3, 3)),
    ("L3 clip 256->256 3d", 16, 32, 32, [256], 256, (3, 3, 3)),
    ("L4 clip 256->256 3d", 16, 16, 16, [256], 256, (3, 3, 3)),
    ("L5 clip 512->512 3d", 16, 8, 8, [512], 512, (3, 3, 3)),
    ("L6 clip 512->512 3d", 16, 4, 4, [512], 512, (3, 3, 3)),
    ("L0 clip 1x1 128->64", 16, 256, 256, [64, 64], 64, (1, 1, 1)),
    ("L1 clip 384->128 2d", 16, 128, 128, [128, 128, 128], 128, (1, 3, 3)),
    ("L2 clip 128->128 2d", 16, 64, 64, [128], 128, (1, 3, 3)),
]


def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] == "bf16" else torch.float32
    tot_f, tot_t = 0.0, 0.0
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, T, H, W, segs, cout, k in SHAPES:
        if only not in name:
            continue
        xs = [torch.randn(T, H, W, c, device=dev).to(dt) for c in segs]
        cin = sum(segs)
        taps = k[0] * k[1] * k[2]
        w = (torch.randn(cout, taps, cin, device=dev) / (taps * cin) ** 0.5).to(dt)
        b = torch.randn(cout, device=dev)
        # FLAIR_BENCH_RES=1: with a residual input (ResBlock conv2), =2: with a per-frame bias (ResBlock conv1: emb)
        mode = int(os.environ.get("FLAIR_BENCH_RES", "0"))
        kw = {}
        if mode == 1:
            kw["res0"] = torch.randn(T, H, W, cout, device=dev).to(dt)
        if mode == 2:
            kw["frame_bias"] = torch.randn(T, cout, device=dev)
        y = ops.conv(xs, w, b, cout, k, **kw)
        torch.cuda.synchronize()
        n = 10
        # replayed from a hipGraph: an eager python loop cannot issue launches faster than ~15 us apart
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(n):
                    ops.conv(xs, w, b, cout, k, out=y, **kw)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        fl = 2.0 * T * H * W * cout * cin * taps
        print(f"{name:24s} {us:9.1f} us  {fl/us/1e6:8.1f} TF/s  ({fl/1e9:7.1f} GF)", flush=True)


if __name__ == "__main__":
    main()
