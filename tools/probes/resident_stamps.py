"""In-kernel s_memtime stamps of conv_resident_kernel (diagnostic build: make -C flair_amd/csrc timing; FLAIR_HIP_LIB=tools/probes/libflair_timing.so):
per wave, the cycles of its 14 stages spent waiting for the stage's DMA, at the barrier, and in the multiply (+ epilogue pieces) phase."""
import os
import torch

dev = torch.device("cuda:0")
stamps = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
os.environ["FLAIR_RES_STAMPS"] = hex(stamps.data_ptr())
os.environ["FLAIR_RES_DEBUG"] = "20"
from flair_amd import ops  # noqa: E402

dt = torch.bfloat16
S, c = 256, 64
x = torch.randn(1, S, S, c, device=dev).to(dt)
w6 = (torch.randn(432, 9, c, device=dev) / (9 * c) ** 0.5).to(dt)
b6 = torch.randn(432, device=dev)
raw = torch.empty(1, S, S, 432, device=dev, dtype=dt)
for act in (4, 0):
    for _ in range(5):
        ops.conv_chain(x, None, None, 0, w6, b6, act, c, 432, out=raw, act_param=10.0 if act == 4 else 0.0, act_period=48 if act == 4 else 0)
    torch.cuda.synchronize()
    nw = int(os.environ.get("FLAIR_CONV_RESIDENT_WAVES", "8"))
    st = stamps[:256 * nw * 4].view(256, nw, 4).double().cpu()
    m = st.mean(dim=(0, 1))
    print(f"act={act}: per wave, memtime ticks summed over 14 stages: wait for DMA {m[0]:8.0f} | barrier {m[1]:8.0f} | multiply phase {m[2]:8.0f} | "
          f"kernel body {m[3]:8.0f}  (per stage: {m[0] / 14:6.0f} / {m[1] / 14:6.0f} / {m[2] / 14:6.0f}); wave 0 vs wave 7 multiply: "
          f"{st[:, 0, 2].mean() / 14:6.0f} / {st[:, nw - 1, 2].mean() / 14:6.0f}")
