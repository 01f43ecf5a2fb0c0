"""Do kernels of two HIP streams run concurrently on this part at all?  Two independent chains of small-footprint launches
(im2col kernel capped at one 256-thread workgroup per CU, 24 KB LDS each: FLAIR_IGEMM_GRID_CAP=256), one per stream, from one
hipGraph: alone vs together.  Also the K-split per-frame kernel (one 512-thread workgroup per CU, ~73 KB LDS)."""
import torch
from flair_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16


def setup(T, H, W, cin, cout):
    x = torch.randn(T, H, W, cin, device=dev).to(dt)
    w = (torch.randn(cout, 9, cin, device=dev) / (9 * cin) ** 0.5).to(dt)
    b = torch.randn(cout, device=dev)
    y = torch.empty(T, H, W, cout, device=dev, dtype=dt)
    return [x, y], w, b


def graph_time(fn, reps=5):
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
    N = 40
    for label, (T, H, W, c) in (("im2col capped (24 KB LDS, 4 waves/CU)", (4, 128, 120, 64)),
                                ("per-frame 128^2 c=128 (K-split kernel, ~73 KB LDS, 8 waves/CU)", (1, 128, 128, 128))):
        A, wA, bA = setup(T, H, W, c, c)
        B, wB, bB = setup(T, H, W, c, c)

        def chain(bufs, w, b):
            for i in range(N):
                ops.conv(bufs[i & 1], w, b, c, (1, 3, 3), act=ops.ACT_LRELU01, out=bufs[(i + 1) & 1])
        t1 = graph_time(lambda: chain(A, wA, bA))
        side = torch.cuda.Stream()

        def both():
            m = torch.cuda.current_stream()
            side.wait_stream(m)
            with torch.cuda.stream(side):
                chain(B, wB, bB)
            chain(A, wA, bA)
            m.wait_stream(side)
        t2 = graph_time(both)
        print(f"{label}: one chain of {N} {t1:8.1f} us | two chains on two streams (one graph) {t2:8.1f} us  -> x{t2 / t1:.2f} "
              f"(1.0 = full overlap, 2.0 = serialised)", flush=True)


if __name__ == "__main__":
    main()
