import torch
from flair_amd import ops
dev = torch.device("cuda:0")
for name, H, c in (("L0 c=64", 256, 64), ("L1 c=128", 128, 128)):
    dt = torch.bfloat16
    x0 = torch.randn(1, H, H, c, device=dev).to(dt); x1 = torch.randn(1, H, H, c, device=dev).to(dt)
    raw = torch.randn(1, H, H, 432, device=dev).to(dt)
    f1 = torch.randn(1, H, H, 2, device=dev) * 2; f2 = torch.randn(1, H, H, 2, device=dev) * 2
    w = (torch.randn(c, 9, 2 * c, device=dev) / (18 * c) ** 0.5).to(dt); b = torch.randn(c, device=dev)
    for _ in range(11):
        ops.dcn_align(x0, x1, raw, f1, f2, w, b, c, groups=16, max_mag=10.0)
    torch.cuda.synchronize()
