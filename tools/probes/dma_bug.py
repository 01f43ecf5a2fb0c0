import torch
from flair_amd import ops
dev = torch.device("cuda:0")
def run(Tb, Hb, Wb, C, cout, tap, co):
    x = torch.randn(Tb, Hb, Wb, C, device=dev, dtype=torch.bfloat16)
    k = 37 % C
    kt, kh, kw = tap
    w3 = torch.zeros(cout, C, 3, 3, 3)
    w3[co, k, kt, kh, kw] = 1.0
    y3 = ops.conv(x, ops.pack_conv_weight(w3, [(C, C)], torch.bfloat16).to(dev), None, cout, (3, 3, 3))
    torch.cuda.synchronize()
    nz = [(y3[..., c] != 0).sum().item() for c in range(cout)]
    print(f"T{Tb} {Hb}x{Wb} C{C} cout{cout} tap{tap} co{co}: nonzero per cout {[n for n in nz if n] or 0} at couts {[i for i,n in enumerate(nz) if n]}", flush=True)
for cout in (8, 16, 32, 64):
    run(5, 256, 256, 64, cout, (2, 0, 1), 5)
run(5, 256, 256, 64, 8, (1, 1, 1), 5)
run(5, 256, 256, 64, 8, (0, 1, 1), 5)
run(5, 256, 256, 64, 8, (1, 1, 1), 0)
run(5, 256, 256, 64, 16, (2, 0, 1), 12)
