"""The c -> 27*G offset convolution of one frame: FLAIR_CONV_RESIDENT=0|1 forms of flair_conv_chain (c = 64, 256^2) and the
LDS-DMA conv kernel (c = 128, 128^2), hipGraph replays of 20 back-to-back launches.  python tools/bench_conv6.py"""
import torch
from flair_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16


def graph_us(fn, n=20):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for S, c in ((256, 64), (128, 128)):
    x = torch.randn(1, S, S, c, device=dev).to(dt)
    w6 = (torch.randn(432, 9, c, device=dev) / (9 * c) ** 0.5).to(dt)
    b6 = torch.randn(432, device=dev)
    raw = torch.empty(1, S, S, 432, device=dev, dtype=dt)
    fl = 2.0 * 9 * c * 432 * S * S
    if c == 64:
        t = graph_us(lambda: ops.conv_chain(x, None, None, 0, w6, b6, 4, c, 432, out=raw, act_param=10.0, act_period=48))
        print(f"{S}^2 c={c} -> 432 (DCN offsets act) flair_conv_chain: {t:6.1f} us  {fl / t / 1e6:6.0f} TFLOP/s")
        t = graph_us(lambda: ops.conv_chain(x, None, None, 0, w6, b6, 0, c, 432, out=raw))
        print(f"{S}^2 c={c} -> 432 (no act)          flair_conv_chain: {t:6.1f} us  {fl / t / 1e6:6.0f} TFLOP/s")
    t = graph_us(lambda: ops.conv(x, w6, b6, 432, (1, 3, 3), act=4, act_param=10.0, act_period=48, out=raw))
    print(f"{S}^2 c={c} -> 432 (DCN offsets act) flair_conv_nhwc : {t:6.1f} us  {fl / t / 1e6:6.0f} TFLOP/s")

# the pair conv_offset[6] -> deformable alignment on one 256^2 frame (c = 64), as the recurrence issues it
S, c, G = 256, 64, 16
x = torch.randn(1, S, S, c, device=dev).to(dt)
p0 = torch.randn(1, S, S, c, device=dev).to(dt)
p1 = torch.randn(1, S, S, c, device=dev).to(dt)
w6 = (torch.randn(432, 9, c, device=dev) / (9 * c) ** 0.5).to(dt)
b6 = torch.randn(432, device=dev)
wd = (torch.randn(c, 9, 2 * c, device=dev) / (18 * c) ** 0.5).to(dt)
bd = torch.randn(c, device=dev)
f1 = torch.randn(1, S, S, 2, device=dev)
f2 = torch.randn(1, S, S, 2, device=dev)
raw = torch.empty(1, S, S, 432, device=dev, dtype=dt)
out = torch.empty(1, S, S, c, device=dev, dtype=dt)


def pair():
    ops.conv_chain(x, None, None, 0, w6, b6, 4, c, 432, out=raw, act_param=10.0, act_period=48)
    ops.dcn_align(p0, p1, raw, f1, f2, wd, bd, c, groups=G, max_mag=10.0, out=out, raw_activated=True)


t = graph_us(pair)
t6 = graph_us(lambda: ops.conv_chain(x, None, None, 0, w6, b6, 4, c, 432, out=raw, act_param=10.0, act_period=48))
td = graph_us(lambda: ops.dcn_align(p0, p1, raw, f1, f2, wd, bd, c, groups=G, max_mag=10.0, out=out, raw_activated=True))
print(f"256^2 c=64: conv_offset[6] + alignment pair {t:6.1f} us  (alone: {t6:5.1f} + {td:5.1f} = {t6 + td:6.1f})")
