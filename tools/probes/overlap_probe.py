"""Can a side stream fill the bubbles of the BasicVSR++ recurrence?  A chain of DEPENDENT per-frame 3x3 convolutions (one
workgroup per CU, latency bound: what the recurrence is made of) on the main stream, and clip-level convolutions with a small
LDS footprint (im2col kernel, 24 KB) or a large one (LDS-DMA kernel, ~150 KB) on a second stream, captured in ONE hipGraph.
Prints: chain alone, side alone, both (ideal: max of the two; serialised: their sum)."""
import sys
import torch
from flair_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16


def conv_setup(T, H, W, cin, cout):
    x = torch.randn(T, H, W, cin, device=dev).to(dt)
    w = (torch.randn(cout, 9, cin, device=dev) / (9 * cin) ** 0.5).to(dt)
    b = torch.randn(cout, device=dev)
    y = torch.empty(T, H, W, cout, device=dev, dtype=dt)
    return x, w, b, y


def timed(fn, reps=5):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def main():
    N = 60
    for (S, c) in ((256, 64), (128, 128)):
        xa, wa, ba, ya = conv_setup(1, S, S, c, c)
        xb = torch.empty_like(xa)

        def chain():
            a_, b_ = xa, xb
            for _ in range(N):
                ops.conv(a_, wa, ba, c, (1, 3, 3), act=ops.ACT_LRELU01, out=b_ if a_ is xa else xa)
                a_, b_ = (b_, a_) if a_ is xa else (xa, xb)
        # keep it simple: ping-pong between two buffers
        bufs = [xa, xb]

        def chain2():
            for i in range(N):
                ops.conv(bufs[i & 1], wa, ba, c, (1, 3, 3), act=ops.ACT_LRELU01, out=bufs[(i + 1) & 1])
        t_chain = timed(chain2)
        for label, Wside in (("im2col kernel, 24 KB LDS (W % 32 != 0)", S - 8), ("LDS-DMA kernel, ~150 KB LDS", S)):
            xs, ws, bs, ys = conv_setup(16, S, Wside, c + 32, c)
            nside = 4

            def side():
                for _ in range(nside):
                    ops.conv(xs, ws, bs, c, (1, 3, 3), out=ys)
            t_side = timed(side)
            side_stream = torch.cuda.Stream()

            def both():
                main_s = torch.cuda.current_stream()
                side_stream.wait_stream(main_s)
                with torch.cuda.stream(side_stream):
                    side()
                chain2()
                main_s.wait_stream(side_stream)
            t_both = timed(both)
            # the same without a graph: eager launches on two streams, events around the pair
            both()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                both()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3)
            t_eager = best
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                chain2()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3)
            t_chain_eager = best
            print(f"{S}x{S} c={c}: chain of {N} per-frame convs {t_chain:8.1f} us ({t_chain / N:.1f} each) | side x{nside} [{label}] "
                  f"{t_side:8.1f} us | both {t_both:8.1f} us  (max {max(t_chain, t_side):.0f}, sum {t_chain + t_side:.0f}) | eager two streams {t_eager:8.1f} us (chain alone eager {t_chain_eager:.0f})", flush=True)


if __name__ == "__main__":
    main()
