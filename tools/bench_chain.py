"""Fused conv chains vs the launches they replace, on the per-frame shapes of the BasicVSR++ recurrence
(config 2: c = 64 at 256x256, c = 128 at 128x128).  Mean over 50 back-to-back launches, HIP events."""
import sys

import torch

from flair_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] == "bf16" else torch.float32


def timeit(fn, n=50):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def w(cout, cin):
    return (torch.randn(cout, 9, cin, device=dev) / (9 * cin) ** 0.5).to(dt)


for name, S, c in (("L0 c=64 256^2", 256, 64), ("L1 c=128 128^2", 128, 128)):
    x = torch.randn(1, S, S, c, device=dev).to(dt)
    r0 = torch.randn(1, S, S, c, device=dev).to(dt)
    r1 = torch.randn(1, S, S, c, device=dev).to(dt)
    wa, wb, w6 = w(c, c), w(c, c), w(432, c)
    ba, bb, b6 = (torch.randn(n, device=dev) for n in (c, c, 432))
    k = (1, 3, 3)
    y = torch.empty_like(x)
    y2 = torch.empty_like(x)
    raw = torch.empty(1, S, S, 432, device=dev, dtype=dt)
    t_a = timeit(lambda: ops.conv(x, wa, ba, c, k, act=2, out=y))
    t_ab = timeit(lambda: (ops.conv(x, wa, ba, c, k, act=2, out=y), ops.conv(y, wb, bb, c, k, act=2, out=y2)))
    t_6 = timeit(lambda: ops.conv(y, w6, b6, 432, k, out=raw))
    t_b6 = timeit(lambda: (ops.conv(x, wb, bb, c, k, act=2, out=y), ops.conv(y, w6, b6, 432, k, out=raw)))
    t_ab6 = timeit(lambda: (ops.conv(x, wa, ba, c, k, act=2, out=y), ops.conv(y, wb, bb, c, k, act=2, out=y2),
                            ops.conv(y2, w6, b6, 432, k, out=raw)))
    c_ab = timeit(lambda: ops.conv_chain(x, wa, ba, 2, wb, bb, 2, c, c, out=y2))
    c_res = timeit(lambda: ops.conv_chain(x, wa, ba, 1, wb, bb, 0, c, c, res0=r0, res1=r1, out_scale=0.9, out=y2))
    c_b6 = timeit(lambda: ops.conv_chain(x, wb, bb, 2, w6, b6, 0, c, 432, out=raw))
    c_6 = timeit(lambda: ops.conv_chain(x, None, None, 0, w6, b6, 0, c, 432, out=raw))
    print(f"{name}: conv c->c {t_a:6.1f} | pair c->c->c: 2 launches {t_ab:6.1f}  chain {c_ab:6.1f} (with 2 residuals {c_res:6.1f}) | "
          f"c->432: conv {t_6:6.1f}  resident-input chain {c_6:6.1f} | c->c->432: 2 launches {t_b6:6.1f}  chain {c_b6:6.1f} | "
          f"c->c->c->432: 3 launches {t_ab6:6.1f}  conv+chain {t_a + c_b6:6.1f}  chain+chain1 {c_ab + c_6:6.1f} us", flush=True)
