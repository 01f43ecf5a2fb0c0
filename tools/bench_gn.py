"""GroupNorm+SiLU (+FiLM) family on the clip tensors of config 2: GB/s of algorithmic bytes (2 reads + 1 write)."""
import torch

from flair_amd import ops

dev = torch.device("cuda:0")
for name, T, S, C in (("L0 16x256^2x64", 16, 256, 64), ("L1 16x128^2x128", 16, 128, 128), ("L2 16x64^2x128", 16, 64, 128),
                      ("L3 16x32^2x256", 16, 32, 256), ("L4 16x16^2x256", 16, 16, 256), ("L4 cat 16x16^2x512", 16, 16, 512),
                      ("L5 16x8^2x512", 16, 8, 512), ("L6 16x4^2x512", 16, 4, 512)):
    x = torch.randn(T, S, S, C, device=dev).to(torch.bfloat16)
    g = torch.ones(C, device=dev)
    b = torch.zeros(C, device=dev)
    film = torch.randn(T, 2 * C, device=dev)
    y = torch.empty_like(x)
    ops.group_norm(x, g, b, act=ops.ACT_SILU, film=film, out=y)
    torch.cuda.synchronize()
    n = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.group_norm(x, g, b, act=ops.ACT_SILU, film=film, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    nbytes = 3 * x.numel() * 2
    print(f"{name}: {us:7.1f} us for the norm = {nbytes / us / 1e3:6.0f} GB/s of 3*numel bytes "
          f"({nbytes / us / 1e3 / 8000:.2f} of 8 TB/s)", flush=True)
