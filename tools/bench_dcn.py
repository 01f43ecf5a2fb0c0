"""Deformable-alignment kernel alone on the per-frame shapes of config 2: pre-activation raw (the kernel derives
10*tanh / sigmoid per gathered group) vs activated raw (FLAIR_ACT_DCN_OFFSETS upstream), same box, same data."""
import torch

from flair_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, H, c in (("L0 c=64 256^2", 256, 64), ("L1 c=128 128^2", 128, 128)):
    dt = torch.bfloat16
    x0 = torch.randn(1, H, H, c, device=dev).to(dt)
    x1 = torch.randn(1, H, H, c, device=dev).to(dt)
    pre = torch.randn(1, H, H, 432, device=dev) * 0.03         # conv_offset's last layer is near zero-init: sub-pixel residues
    ch = torch.arange(432, device=dev)
    act = torch.where((ch % 48) < 32, 10 * torch.tanh(pre), torch.sigmoid(pre)).to(dt)
    pre = pre.to(dt)
    smooth = lambda: torch.nn.functional.interpolate(torch.randn(1, 2, H // 16, H // 16, device=dev), size=(H, H),  # noqa: E731
                                                     mode="bilinear").permute(0, 2, 3, 1).contiguous()
    f1, f2 = smooth(), smooth() * 2                            # smooth optical-flow-like fields, about a pixel
    w = (torch.randn(c, 9, 2 * c, device=dev) / (18 * c) ** 0.5).to(dt)
    b = torch.randn(c, device=dev)
    t0 = timeit(lambda: ops.dcn_align(x0, x1, pre, f1, f2, w, b, c, groups=16, max_mag=10.0))
    t1 = timeit(lambda: ops.dcn_align(x0, x1, act, f1, f2, w, b, c, groups=16, max_mag=10.0, raw_activated=True))
    print(f"{name}: dcn_align pre-activation raw {t0:6.1f} us | activated raw {t1:6.1f} us", flush=True)
