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
    # demo hyper-parameters of scripts/video_sample.py:499-556
    "gaussian": dict(w=0.75, rho=0.25, noise_level=2.55, zeta=1.0, jpeg_qf=-1, factor=4),
    "jpeg": dict(w=0.5, rho=0.5, noise_level=12.75, zeta=1.0, jpeg_qf=60, factor=4),
    "x8_bicubic": dict(w=0.85, rho=0.85, noise_level=0.0, zeta=-1, jpeg_qf=-1, factor=8, face_weight=0.93),
    "x16_bicubic": dict(w=0.7, rho=0.85, noise_level=0.0, zeta=-1, jpeg_qf=-1, factor=16, face_weight=0.98),
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


def clip_inputs(task, clip_id, frames, size, lr_factor=None):
    """Host tensors of one synthetic clip: degraded (1,T,3,s,s) in [-1,1], init (1,T,3,S,S) in
    [-1,1] (INIT_FUNC of scripts/video_sample.py:158-163: area resize for the blur tasks, bicubic
    for the bicubic tasks), rnn_input (1,T,3,S,S)."""
    g = torch.Generator().manual_seed(1234 + clip_id)
    s = size // (lr_factor or TASKS[task]["factor"])
    degraded = torch.rand(1, frames, 3, s, s, generator=g)
    if "bicubic" in task:
        init = F.interpolate(degraded[0], (size, size), mode="bicubic", align_corners=False).clamp(0, 1)[None]
    else:
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


def codeformer_aux(gan):
    """The reference's ``aux_model`` closure (scripts/video_sample.py:450-452) around a CodeFormer prior: the
    sampler hands it pred_xstart of aligned 512x512 faces and blends its first output back in."""
    def aux_model(x0, *args, **kwargs):
        return gan(x0, w=1.0, adain=True)[0]
    return aux_model


# ------------------------------------------------------------------ bicubic tasks (sr3.UNet)
def sr3_config(image_size, use_fp16=True):
    """MODEL_CONFIG['x8_bicubic'] of scripts/video_sample.py:78-96 at clip side `image_size`
    (attention / BasicVSR++ resolutions scale with it: SURVEY.md section 8d)."""
    return dict(image_size=image_size, in_channel=6, out_channel=3, inner_channel=64, norm_groups=16,
                channel_mults=(1, 2, 4, 8, 16), attn_res=(image_size // 8, image_size // 16),
                vsrpp_res=(image_size, image_size // 2), spatial_attn=False, temporal_attn=True, res_blocks=1,
                dropout=0.0, dtype=torch.bfloat16 if use_fp16 else torch.float32, cross_frame_module=True,
                use_checkpoint=True, num_frames=7, head_dim=64)


def bicubic_diffusion_for(steps):
    from .guided_diffusion.script_util import create_gaussian_diffusion
    return create_gaussian_diffusion(diffusion_steps=2000, learn_sigma=False, noise_schedule="face_bicubic",
                                     timestep_respacing=str(steps))


def bicubic_taps(factor, a=-0.5):
    """The 4*factor-tap bicubic kernel scripts/video_sample.py:205-226 hands to SRConv."""
    def k(x):
        x = abs(x)
        if x <= 1:
            return (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1
        if x < 2:
            return a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a
        return 0.0
    n = factor * 4
    taps = np.array([k((1 / factor) * (i - np.floor(n / 2) + 0.5)) for i in range(n)])
    taps = taps / taps.sum()
    t = torch.from_numpy(taps).float()
    return t / t.sum()


def parsenet_weights_fn(parser, task):
    """``vsrpp_weights_fn`` for flair_amd.video built on a ParseNet (flair_amd.guided_diffusion.parsenet): the
    per-pixel propagation weights of scripts/video_sample.py:427-444 -- background (parsing class 0) pixels get
    TASKS[task]['face_weight'] (0.93 for x8, 0.98 for x16 bicubic), the rest 1 -- as (1, T, 1, S, S)."""
    w_face = TASKS[task]["face_weight"]

    def fn(init_norm):                               # (1, T, 3, S, S) in [-1, 1]
        return parser.face_weight(init_norm[0].float().contiguous(), w_face)[None]
    return fn


def face_weight_map(frames, size, inside):
    """Stand-in for the face-parsing mask of scripts/video_sample.py:427-444: an ellipse of
    background (mask=1 -> weight `inside`... the script weights background pixels) per frame."""
    ys = torch.linspace(-1, 1, size).view(size, 1)
    xs = torch.linspace(-1, 1, size).view(1, size)
    face = ((xs / 0.55) ** 2 + (ys / 0.75) ** 2 <= 1).float()
    background = 1 - face
    w = background * inside + (1 - background) * 1.0
    return w.view(1, 1, 1, size, size).repeat(1, frames, 1, 1, 1)
