"""Primitive layer containers + helpers with the reference's names (guided_diffusion/nn_new.py).

In this package ``torch.nn`` modules are *parameter containers only*: they give the
model the reference's state-dict names, shapes and default initialisation, but their
``forward`` is never called -- all arithmetic happens in libflair_hip.so (see
``flair_amd/ops.py``).  ``timestep_embedding`` is the one function of the reference's
nn_new.py that scripts call directly; it runs the HIP kernel.
"""
import torch
import torch.nn as nn

from .. import ops


class GroupNorm32(nn.GroupNorm):
    """Parameter container for the fp32 GroupNorm of the reference (nn_new.py:17-19)."""

    def forward(self, x):  # pragma: no cover - never on the product path
        raise RuntimeError("flair_amd layers are executed by the HIP engine, not by torch")


def conv_nd(dims, *args, **kwargs):
    """nn_new.py:22-32."""
    if dims == 1:
        return nn.Conv1d(*args, **kwargs)
    if dims == 2:
        return nn.Conv2d(*args, **kwargs)
    if dims == 3:
        return nn.Conv3d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def normalization(channels):
    """nn_new.py:93-100."""
    return GroupNorm32(32, channels)


def zero_module(module):
    """nn_new.py:68-74."""
    for p in module.parameters():
        p.detach().zero_()
    return module


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal embeddings (nn_new.py:103-121) on the GPU; timesteps: 1-D device tensor."""
    return ops.timestep_embedding(timesteps.float().contiguous(), dim, float(max_period))
