"""Sliding-window restoration of a whole video: the caller side of the sampling hot path.

Mirror of the window loop of the reference's ``scripts/video_sample.py:334-492`` (SURVEY.md section
8f, "next" row 2) without its file I/O: the frames are cut into windows of ``FRAME_SLICE_LEN`` with
``OVERLAP`` frames shared between neighbours (``more_itertools.windowed`` semantics,
video_sample.py:361-368), every window is initialised by the task's ``INIT_FUNC`` (:158-163),
sampled with ``diffusion.sample`` and stitched: all but the first window pin their first ``OVERLAP``
frames to the previous window's result (``prev_recon``, gaussian_diffusion.py:497-505) and
contribute only the frames after them (:481-485).

Everything tensor-valued runs through the HIP kernels (resize / clamp / sampler); the face-parsing
weights and the CodeFormer prior are passed in as callables (``vsrpp_weights_fn`` / ``aux_model``; the HIP
versions are ``workload.parsenet_weights_fn(ParseNet(...), task)`` and ``workload.codeformer_aux(CodeFormer(...))``).
"""
import torch

from . import ops
from . import workload as wl

FRAME_SLICE_LEN = 10   # scripts/video_sample.py:202
OVERLAP = 3            # scripts/video_sample.py:203


def window_indices(n_frames, length=FRAME_SLICE_LEN, overlap=OVERLAP):
    """Frame indices of every window: ``more_itertools.windowed(range(n), length, step=length-overlap)``
    with the ``None`` padding of the last window dropped (video_sample.py:361-368)."""
    if length <= 0 or not 0 <= overlap < length:
        raise ValueError("need 0 <= overlap < length")
    step = length - overlap
    if n_frames <= 0:
        return []
    if n_frames <= length:
        return [list(range(n_frames))]
    return [list(range(s, min(s + length, n_frames))) for s in range(0, n_frames - length + step, step)]


_consts = {}


def _const(dev, value):
    """Cached 4-element device constant for flair_affine_channels_f32's per-channel terms."""
    key = (dev, float(value))
    if key not in _consts:
        _consts[key] = torch.full((4,), float(value), device=dev)
    return _consts[key]


def _to_clip(x):
    N, C, h, w = x.shape
    clip = torch.zeros((N, h, w, 4), dtype=torch.float32, device=x.device)
    return ops.nchw_to_clip(x.float().contiguous(), clip, 0)


def _affine(clip, a, b, lo, hi, sub=0.0, mul=1.0):
    """(clamp(x*a + b, lo, hi) - sub) * mul on the 3 image channels of an f32 clip tensor -> (N,3,H,W)."""
    out = torch.zeros_like(clip)
    ops.affine_channels(clip, 3, a, b, lo, hi, _const(clip.device, sub), _const(clip.device, mul), out)
    return ops.clip_to_nchw(out, 3)


def init_frames(task, degraded01, size):
    """INIT_FUNC (video_sample.py:158-163) followed by the (x - 0.5) / 0.5 of :372-373: bicubic for the
    bicubic tasks, ``area`` (= block replication when upscaling by an integer factor) for the blur tasks,
    clamped to [0, 1], returned in [-1, 1]."""
    clip = _to_clip(degraded01)
    if "bicubic" in task:
        mode = ops.RESIZE_BICUBIC
    else:
        if size % degraded01.shape[-1] or size % degraded01.shape[-2]:
            raise ValueError("area initialisation needs an integer upscaling factor")
        mode = ops.RESIZE_NEAREST
    big = ops.resize(clip, (size, size), mode, channels=3)
    return _affine(big, 1.0, 0.0, 0.0, 1.0, sub=0.5, mul=2.0)


def normalise(degraded01):
    """(x - 0.5) / 0.5 (video_sample.py:372); returns the NCHW images and their clip form."""
    clip = _to_clip(degraded01)
    out = torch.zeros_like(clip)
    ops.affine_channels(clip, 3, 2.0, -1.0, float("-inf"), float("inf"), _const(clip.device, 0.0),
                        _const(clip.device, 1.0), out)
    return ops.clip_to_nchw(out, 3), out


def rnn_input(degraded_norm_clip, size):
    """The flow-network input of the blur tasks (video_sample.py:406-425): the two ``VF.normalize`` calls
    cancel around the bicubic resize, what remains is resize + clamp to [-1, 1]."""
    big = ops.resize(degraded_norm_clip, (size, size), ops.RESIZE_BICUBIC, channels=3)
    return _affine(big, 1.0, 0.0, -1.0, 1.0)


def restore_window(task, degraded01, model, diffusion, restore_fn_for, *, size, prev_recon=None, overlap=OVERLAP,
                   window_index=0, aux_model=wl.identity_aux, vsrpp_weights_fn=None, hp=None, tau=5, t_start=-1,
                   noise_fn=None, q_noise_fn=None):
    """One window of the loop (video_sample.py:371-485).  degraded01: (1, T, 3, h, w) in [0, 1] on the GPU;
    prev_recon: the previous window's last ``overlap`` results ((1, <=overlap, 3, S, S), [-1, 1] domain) or None.
    Returns (frames01 of the frames this window contributes, (T', 3, S, S) in [0, 1]; next prev_recon)."""
    hp = hp or wl.TASKS[task]
    dev = degraded01.device
    wi = window_index
    deg = degraded01[0].float().contiguous()                                      # (T,3,h,w) in [0,1]
    T = deg.shape[0]
    init_n = init_frames(task, deg, size)[None]                                   # (1,T,3,S,S) in [-1,1]
    deg_n, deg_n_clip = normalise(deg)
    deg_n = deg_n[None]
    t0 = diffusion.num_timesteps - 1 if t_start == -1 else t_start
    tt = torch.full((T,), t0, device=dev, dtype=torch.long)
    qn = q_noise_fn(wi, init_n[0]) if q_noise_fn is not None else None
    noise = diffusion.q_sample(init_n[0].contiguous(), tt, noise=qn)
    kwargs = dict(low_res_input=init_n, num_frames=T, enable_cross_frames=True,
                  vsrpp_weights=vsrpp_weights_fn(init_n) if vsrpp_weights_fn is not None else 1.0)
    if "bicubic" not in task:
        kwargs["rnn_input"] = rnn_input(deg_n_clip, size)[None]
    sample = diffusion.sample(
        model, noise, model_kwargs=kwargs, device=dev, progress=False, clip_denoised=True,
        restore_fn=restore_fn_for(deg_n), post_fn=None, face_restore_helper=None, aux_model=aux_model,
        w=hp["w"], tau=tau, affine_matrices=None, aligned=True, sample_mode="ddpm", rho=hp["rho"],
        noise_level=hp["noise_level"], prev_recon=prev_recon, zeta=hp["zeta"], t_start=t_start,
        noise_fn=(lambda it, like, _wi=wi: noise_fn(_wi, it, like)) if noise_fn is not None else None)
    keep = sample if prev_recon is None else sample[overlap:]                    # (T',3,S,S), [-1,1] domain
    nxt = keep[-overlap:].clone()[None] if overlap > 0 else None                 # (1,<=overlap,3,S,S), :481-483
    frames01 = _affine(_to_clip(keep.contiguous()), 0.5, 0.5, 0.0, 1.0)          # (clamp(x,-1,1)+1)/2
    return frames01, nxt


def restore_video(task, degraded01, model, diffusion, restore_fn_for, *, size, aux_model=wl.identity_aux,
                  vsrpp_weights_fn=None, hp=None, tau=5, t_start=-1, length=FRAME_SLICE_LEN, overlap=OVERLAP,
                  noise_fn=None, q_noise_fn=None):
    """degraded01: (1, N, 3, h, w) frames in [0, 1] on the GPU.  Returns (N, 3, size, size) in [0, 1].

    restore_fn_for(degraded_norm_window (1,T,3,h,w)) -> restore_fn(x0) is the data-consistency
    operator of the window (video_sample.py:455-459); vsrpp_weights_fn(init_norm (1,T,3,S,S)) supplies
    the per-pixel propagation weights of the bicubic tasks (face parsing, :427-444) and defaults to 1.0;
    noise_fn / q_noise_fn(window_index, like) let tests share one noise tape with the oracle.
    (File-to-file form with decode / upload / encode overlapped: flair_amd.io.restore_video_files.)"""
    dev = degraded01.device
    n_frames = degraded01.shape[1]
    prev_recon = None
    out = torch.empty((n_frames, 3, size, size), dtype=torch.float32, device=dev)
    filled = 0
    for wi, idx in enumerate(window_indices(n_frames, length, overlap)):
        frames01, prev_recon = restore_window(
            task, degraded01[:, idx[0]:idx[-1] + 1], model, diffusion, restore_fn_for, size=size,
            prev_recon=prev_recon, overlap=overlap, window_index=wi, aux_model=aux_model,
            vsrpp_weights_fn=vsrpp_weights_fn, hp=hp, tau=tau, t_start=t_start, noise_fn=noise_fn, q_noise_fn=q_noise_fn)
        out[filled:filled + frames01.shape[0]].copy_(frames01)
        filled += frames01.shape[0]
    return out[:filled]
