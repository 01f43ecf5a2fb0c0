"""FLOPs of one UNet forward of the benched layout, counted on the CPU oracle with torch.utils.flop_counter at 8 x 128^2 and
scaled to 16 x 256^2 (convolutions / linear-on-pixels scale with T*H*W = x8; spatial attention QK^T / AV with
frames * L^2 = x2 * 16 = x32).  Usage: python tools/count_flops.py [--attention-resolutions 16,32,64]
Test infrastructure: imports oracle/ (never used by the product path)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--attention-resolutions", default=None)
    a = ap.parse_args()
    from torch.utils.flop_counter import FlopCounterMode
    from flair_amd import workload as wl
    from oracle.unet import UNetModel as Oracle
    S, T = a.size, a.frames
    cfg = wl.blur_config(S, use_fp16=False)
    if a.attention_resolutions:
        # given for the 256-pixel bench: keep the same downsample rates at the probe size
        cfg["attention_resolutions"] = tuple(int(v) for v in a.attention_resolutions.split(","))
    torch.manual_seed(0)
    m = Oracle(**cfg).eval()
    n_params = sum(p.numel() for p in m.parameters())
    degraded, init, rnn = wl.clip_inputs("gaussian", 0, T, S)
    x = torch.randn(T, 3, S, S)
    t = torch.full((T,), 10, dtype=torch.long)
    with torch.no_grad(), FlopCounterMode(display=False, depth=2) as fc:
        m(x, t, low_res_input=init, num_frames=T, rnn_input=rnn, vsrpp_weights=1.0)
    by_mod = fc.get_flop_counts()
    total = sum(by_mod["Global"].values())
    spynet = sum(v for k, d in by_mod.items() if "spynet" in k.lower() for v in d.values() if k.count(".") <= 1)
    ops = {str(k): v for k, v in by_mod["Global"].items()}
    out = {"size": S, "frames": T, "attention_resolutions": list(cfg["attention_resolutions"]), "params_M": n_params / 1e6,
           "total_TFLOP": total / 1e12, "spynet_TFLOP": spynet / 1e12, "by_op_TFLOP": {k: v / 1e12 for k, v in ops.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
