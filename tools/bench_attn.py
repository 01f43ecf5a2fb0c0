"""Isolated measurement of the spatial QKVAttention kernel (north_star: >= 50 % MFMA utilisation target).

    python tools/bench_attn.py [--json out.json]

Shapes: the attention blocks of the 16-frame clips -- L = 256 tokens (256x256 clip at ds16; 512x512 at ds32),
L = 1024 (512x512 at ds16), L = 64; 16 frames x heads of width 64.  FLOPs = 4 * frames * heads * L^2 * 64
(QK^T + AV) over the mean launch time of 200 back-to-back launches bracketed by HIP events on the
launch stream; peak = 2.5 PFLOP/s dense bf16.  Also prints the grid (the kernel launches
(ceil(L/128), frames*heads) workgroups of 4 waves), which is what bounds it at these sizes.
"""
import json
import sys

import torch

from flair_amd import ops

PEAK_TFLOPS = 2500.0
SHAPES = [  # name, frames, L (= H*W), C, heads
    ("unet_new ds16 @256^2: L=256, C=256, 4 heads", 16, 256, 256, 4),
    ("unet_new ds32 @256^2: L=64, C=512, 8 heads", 16, 64, 512, 8),
    ("ds16 @512^2: L=1024, C=256, 4 heads", 16, 1024, 256, 4),
    ("ds16 @512^2, 32 frames: L=1024, C=256, 4 heads", 32, 1024, 256, 4),
    ("L=4096 (ds8 @512^2), C=128, 2 heads", 16, 4096, 128, 2),
]


def main():
    dev = torch.device("cuda:0")
    rows = []
    for name, F_, L, C, heads in SHAPES:
        side = int(L ** 0.5)
        qkv = torch.randn(F_, side, side, 3 * C, device=dev).to(torch.bfloat16)
        out = ops.qkv_attention(qkv, heads)
        torch.cuda.synchronize()
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.qkv_attention(qkv, heads, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        flops = 4.0 * F_ * heads * L * L * 64
        tf = flops / us / 1e6
        rows.append({"shape": name, "frames": F_, "L": L, "heads": heads, "workgroups": ((L + 127) // 128) * F_ * heads if ((L + 127) // 128) * F_ * heads >= 256 else ((L + 63) // 64) * F_ * heads,
                     "us_per_launch": us, "GFLOP": flops / 1e9, "TFLOP_s": tf, "frac_of_2.5PF": tf / PEAK_TFLOPS})
        print(f"{name:52s} {us:8.1f} us  {tf:8.1f} TFLOP/s = {tf / PEAK_TFLOPS:6.3f} of peak "
              f"({rows[-1]['workgroups']} workgroups)", flush=True)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump({"kernel": "attn_mfma_bf16_kernel", "peak_TFLOP_s": PEAK_TFLOPS, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
