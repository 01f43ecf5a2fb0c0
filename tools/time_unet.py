"""Time one UNet forward of the BASELINE config-2 model (random init) on the GPU."""
import argparse
import time

import torch

from flair_amd.guided_diffusion.unet_new import UNetModel


def full_config(image_size):
    return dict(image_size=image_size, in_channels=6, model_channels=128, out_channels=6, num_res_blocks=2,
                attention_resolutions=(image_size // 32, image_size // 16, image_size // 8),
                rnn_resolutions=(1, 2), channel_mult=(0.5, 1, 1, 2, 2, 4, 4), use_fp16=True,
                num_head_channels=64, resblock_updown=True, use_scale_shift_norm=True,
                temporal_block=True, use_checkpoint=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    t0 = time.time()
    m = UNetModel(**full_config(a.size))
    with torch.no_grad():
        for p in m.parameters():
            if p.abs().sum() == 0:
                p.normal_(0, 0.02)
    m = m.to(dev).eval()
    if a.dtype == "f32":
        m.convert_to_fp32()
    print(f"build {time.time()-t0:.1f}s params {sum(p.numel() for p in m.parameters())/1e6:.1f}M", flush=True)
    T, S = a.frames, a.size
    x = torch.randn(T, 3, S, S, device=dev)
    lr = (torch.rand(1, T, 3, S, S, device=dev) * 2 - 1)
    t = torch.full((T,), 500, device=dev, dtype=torch.long)
    kw = dict(low_res_input=lr, num_frames=T, vsrpp_weights=1.0)
    t0 = time.time()
    y = m(x, t, **kw)
    torch.cuda.synchronize()
    print(f"first forward (pack + spynet) {time.time()-t0:.2f}s  out {tuple(y.shape)} finite={torch.isfinite(y).all().item()}", flush=True)
    if a.graph:
        m.enable_hip_graph()
        t0 = time.time()
        y = m(x, t, **kw)
        torch.cuda.synchronize()
        print(f"graph capture {time.time()-t0:.2f}s", flush=True)
    for _ in range(a.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.time()
        e0.record()
        y = m(x, t, **kw)
        e1.record()
        torch.cuda.synchronize()
        print(f"forward: gpu {e0.elapsed_time(e1):.1f} ms  wall {1e3*(time.time()-t0):.1f} ms", flush=True)
    print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)


if __name__ == "__main__":
    main()
