"""Per-phase cycles of conv_chain_kernel from in-kernel s_memtime stamps (diagnostic build:
`make -C flair_amd/csrc probe` -> tools/probes/libchain_probe.so; the product library has no stamps)."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from flair_amd import _lib, ops  # noqa: E402

probe = ctypes.CDLL(os.path.join(HERE, "libchain_probe.so"))
probe.flair_last_error.restype = ctypes.c_char_p
_lib._lib = probe          # route ops.conv_chain through the stamped build
dev = torch.device("cuda:0")
dt = torch.bfloat16
NAMES = ["setup + first fetch + LDS write", "stage A K loop", "intermediate -> LDS", "stage B K loop (batch 0)",
         "epilogue (batch 0)", "remaining batches"]


def run(name, S, c, hasA, coutB):
    x = torch.randn(1, S, S, c, device=dev).to(dt)
    wa = (torch.randn(c, 9, c, device=dev) / (9 * c) ** 0.5).to(dt) if hasA else None
    wb = (torch.randn(coutB, 9, c, device=dev) / (9 * c) ** 0.5).to(dt)
    ba = torch.randn(c, device=dev) if hasA else None
    bb = torch.randn(coutB, device=dev)
    dbg = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
    os.environ["FLAIR_CHAIN_DBG_PTR"] = hex(dbg.data_ptr())
    for _ in range(5):
        y = ops.conv_chain(x, wa, ba, 2, wb, bb, 2, c, coutB)
    torch.cuda.synchronize()
    nwg = int((dbg.view(-1, 16)[:, 0] != 0).sum().item())
    st = dbg.view(-1, 16)[:nwg].cpu().double()
    d = (st[:, 1:7] - st[:, 0:6]).mean(0)
    tot = (st[:, 6] - st[:, 0]).mean().item()
    print(f"{name}: {nwg} workgroups, mean workgroup lifetime {tot:.0f} cycles")
    if not hasA:                       # stamps 1, 2 are not written: 0 -> 3 is the input staging
        d[0] = (st[:, 3] - st[:, 0]).mean()
    for k in range(6):
        if hasA or k not in (1, 2):
            print(f"    {NAMES[k]:34s} {d[k].item():9.0f} cycles")


run("L0 pair 64->64->64 @256^2", 256, 64, True, 64)
run("L1 pair 128->128->128 @128^2", 128, 128, True, 128)
run("L0 resident-input 64->432", 256, 64, False, 432)
run("L1 resident-input 128->432", 128, 128, False, 432)
