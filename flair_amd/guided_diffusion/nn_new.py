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


class SiLU(nn.SiLU):
    """nn_new.py:12-14 (layout marker: the activation is fused into the GroupNorm / conv kernels)."""


class AvgPool2x2(nn.Module):
    """What ``avg_pool_nd(2, kernel_size=2, stride=2)`` returns here: a marker whose ``run``
    executes the 2x2 average pool of ``flair_resize_nhwc`` on an NHWC clip tensor."""

    def __init__(self, kernel_size=2, stride=None):
        super().__init__()
        stride = kernel_size if stride is None else stride
        if (kernel_size, stride) not in ((2, 2), ((2, 2), (2, 2))):
            raise NotImplementedError("flair_amd: only the 2x2 / stride-2 average pool FLAIR uses is implemented")

    def run(self, x):
        T, H, W, _ = x.shape
        return ops.resize(x, (H // 2, W // 2), ops.RESIZE_AVGPOOL2)

    def forward(self, x):  # pragma: no cover - never on the product path
        raise RuntimeError("flair_amd layers are executed by the HIP engine, not by torch")


def avg_pool_nd(dims, *args, **kwargs):
    """nn_new.py:42-52.  FLAIR only builds the 2-D 2x2 pool (unet_new.py:186-191)."""
    if dims == 2:
        return AvgPool2x2(*args, **kwargs)
    raise ValueError(f"flair_amd: avg_pool_nd supports dims=2 only (got {dims})")


def scale_module(module, scale):
    """nn_new.py:77-83."""
    for p in module.parameters():
        p.detach().mul_(scale)
    return module


def checkpoint(func, inputs, params, flag):
    """nn_new.py:124-140.  Gradient checkpointing trades backward-pass memory for recompute; this
    package only samples (no autograd graph is ever built), so both settings of ``flag`` evaluate
    ``func(*inputs)`` exactly as the reference's ``CheckpointFunction.forward`` does under
    ``no_grad`` (nn_new.py:142-150)."""
    return func(*inputs)
