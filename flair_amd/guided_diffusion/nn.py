"""Names of the reference's ``guided_diffusion/nn.py`` that its model files import
(unet_new.py:19, unet.py:14-27, sr3.py:21-31), bound to this package's containers.

The reference's ``LazyReshaper2D/3D`` (nn.py:350-367) transpose the whole activation around every
conv / GroupNorm; here activations are NHWC clip tensors and those wrappers only keep the
``wrapped_module`` level of the parameter names.  ``conv_nd`` / ``normalization`` of *this* module
wrap their result like the reference's nn.py:751-761,842-849 do (the nn_new.py variants do not).
The tiling / CPU-offload helpers (``SliceProcessor*``, ``patchify``; nn.py:18-338,397-581) are unused
by every shipped configuration and are not provided: a 16x256x256 clip peaks at 3.8 GB of 288 GB.
"""
from .nn_new import (GroupNorm32, SiLU, avg_pool_nd, checkpoint, linear, scale_module,  # noqa: F401
                     timestep_embedding, zero_module)
from .nn_new import conv_nd as _conv_nd
from .nn_new import normalization as _normalization
from .unet_new import FalshAttn, LazyReshaper2D, LazyReshaper3D, PlaceHolder  # noqa: F401


def conv_nd(dims, *args, **kwargs):
    """nn.py:751-761: 2-D / 3-D convolutions come wrapped (parameter names gain ``wrapped_module``)."""
    m = _conv_nd(dims, *args, **kwargs)
    if dims == 2:
        return LazyReshaper2D(m)
    if dims == 3:
        return LazyReshaper3D(m)
    return m


def normalization(channels, *args, **kwargs):
    """nn.py:842-849."""
    return LazyReshaper3D(_normalization(channels))


def flash_attn_wrapper(q, k, v, dropout):
    """nn.py:370-386: ``flash_attn_func`` on (B, L, heads, D) tensors after a cast to fp16, result cast back --
    exact softmax(q k^T / sqrt(D)) v per head.  TemporalAttention itself runs on the fused window-attention kernel
    (``flair_amd.ops.temporal_attention``, which consumes the packed q|k|v clip tensor); this entry keeps the reference's
    calling convention for other callers: q | k | v are packed into one (B, 1, L, 3*heads*D) clip tensor and handed to
    ``flair_attention_wide`` (inputs rounded through fp16 like the reference's cast; dropout must be 0: inference)."""
    import torch
    from .. import ops
    if dropout:
        raise NotImplementedError("flair_amd: flash_attn_wrapper is an inference entry (dropout must be 0)")
    if not q.is_cuda:
        raise RuntimeError("flair_amd.nn.flash_attn_wrapper runs on the MI355X only (tensors must be on 'cuda')")
    B, L, Hh, D = q.shape
    dt = q.dtype
    work = torch.bfloat16 if dt == torch.bfloat16 else torch.float32
    parts = [t.contiguous().to(torch.float16).to(work).reshape(B, 1, L, Hh * D) for t in (q, k, v)]
    qkv = torch.cat(parts, dim=-1).contiguous()
    C = Hh * D
    out = ops.attention_wide(qkv, Hh, D, q_off=0, k_off=C, v_off=2 * C, head_stride=D)
    return out.reshape(B, L, Hh, D).to(dt)
