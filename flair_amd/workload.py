"""Synthetic FLAIR workloads shared by bench.py, __graft_entry__.smoke() and the tests.

Inputs follow SURVEY.md section 8(d): ``degraded = rand(1,T,3,h,w)`` (seed 1234 + clip id),
``init`` = area-resize to SxS, ``x_T = q_sample(init, T-1)`` with seed 4321 + clip id,
random-init weights from ``torch.manual_seed(0)`` with zero-initialised modules re-drawn
N(0, 0.02), ``aux_model`` = identity with ``aligned=True`` (no face library), demo
hyper-parameters of scripts/video_sample.py:499-556.  Everything here is host-side
setup (the caller's side of the hot path); the timed work is in the sampler/UNet.
"""
import numpy as np
import torch
import torch.nn.functional as F

TASKS = {
    # task: (w, rho, noise_level, zeta, jpeg_qf)
    "gaussian": dict(w=0.75, rho=0.25, noise_level=2.55, zeta=1.0, jpeg_qf=-1),
    "jpeg": dict(w=0.5, rho=0.5, noise_level=12.75, zeta=1.0, jpeg_qf=60),
}


def synthetic_blur_kernel(size=25, sigma=2.5):
    """Stand-in for miscs/kernels_12.mat['kernels'][0,3] (a 25x25 f32 blur kernel, sum 1)."""
    r = np.arange(size) - size // 2
    g = np.exp(-0.5 * (r / sigma) ** 2)
    k = np.outer(g, g)
    k = (k / k.sum()).astype(np.float32)
    return (k / k.sum()).astype(np.float32)


def randomize_zero_modules(model, seed=1):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            if p.abs().sum() == 0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)


def clip_inputs(task, clip_id, frames, size, lr_factor=4):
    """Host tensors of one synthetic clip: degraded (1,T,3,s,s) in [-1,1], init (1,T,3,S,S) in
    [-1,1], rnn_input (1,T,3,S,S), noise seed."""
    g = torch.Generator().manual_seed(1234 + clip_id)
    s = size // lr_factor
    degraded = torch.rand(1, frames, 3, s, s, generator=g)
    init = F.interpolate(degraded[0], (size, size), mode="area").clamp(0, 1)[None]
    degraded_n = (degraded - 0.5) / 0.5
    init_n = (init - 0.5) / 0.5
    rnn = F.interpolate(degraded_n[0], (size, size), mode="bicubic", align_corners=False).clamp(-1, 1)[None]
    return degraded_n, init_n, rnn


def blur_config(image_size, use_fp16=True):
    from .guided_diffusion.script_util import blur_unet_config
    return blur_unet_config(image_size, use_fp16=use_fp16)


def diffusion_for(steps, learn_sigma=True):
    from .guided_diffusion.script_util import create_gaussian_diffusion
    return create_gaussian_diffusion(diffusion_steps=1000, learn_sigma=learn_sigma, noise_schedule="face_blur",
                                     timestep_respacing=str(steps), rescale_learned_sigmas=True)


def identity_aux(x0, t, xt):
    return x0
