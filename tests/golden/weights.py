"""Deterministic weights derived from parameter NAMES (shared by fixture generation and tests).

Every parameter / buffer named ``n`` of shape ``s`` is filled from a generator seeded with
crc32(n): conv / linear weights ~ N(0, 1/fan_in) (so activations stay O(1) through the net),
norm scales ~ 1 + 0.1 N, biases ~ 0.05 N.  Zero-initialised modules of the reference get
non-zero values too, so no branch is hidden.  Registered buffers (SPyNet mean/std) are kept, except
BatchNorm running statistics, which are name-seeded as well.
"""
import math
import zlib

import torch


def name_seeded_weights(model):
    with torch.no_grad():
        for name, p in model.named_parameters():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if p.dim() >= 2:
                fan_in = p[0].numel()
                v = torch.randn(p.shape, generator=g) / math.sqrt(fan_in)
            elif name.endswith("weight"):          # norm scale
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            else:
                v = 0.05 * torch.randn(p.shape, generator=g)
            p.copy_(v.to(p.dtype))
        # BatchNorm running statistics (ParseNet): non-trivial values, variances bounded away from zero
        for name, b in model.named_buffers():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.6 + 0.8 * torch.rand(b.shape, generator=g))
    return model
