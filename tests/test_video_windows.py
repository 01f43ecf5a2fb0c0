"""Sliding-window video harness (flair_amd/video.py; SURVEY.md 8f "next" row 2)."""
import pytest
import torch
import torch.nn.functional as F


def _windowed_reference(seq, n, step):
    """Restatement of more_itertools.windowed(seq, n, step=step) from its documented behaviour
    (more-itertools 10.x docs: windows of length n every `step` items, the last one padded with None
    when the items do not divide evenly; a single padded window when len(seq) < n)."""
    seq = list(seq)
    if len(seq) < n:
        return [tuple(seq + [None] * (n - len(seq)))]
    out, i = [], 0
    while i + n <= len(seq):
        out.append(tuple(seq[i:i + n]))
        i += step
    if (len(seq) - n) % step:
        tail = seq[i:]
        out.append(tuple(tail + [None] * (n - len(tail))))
    return out


def test_windowed_reference_matches_documented_examples():
    assert _windowed_reference([1, 2, 3, 4, 5], 3, 1) == [(1, 2, 3), (2, 3, 4), (3, 4, 5)]
    assert _windowed_reference([1, 2, 3], 4, 1) == [(1, 2, 3, None)]
    assert _windowed_reference([1, 2, 3, 4, 5, 6], 3, 2) == [(1, 2, 3), (3, 4, 5), (5, 6, None)]
    assert _windowed_reference([1, 2, 3, 4, 5, 6, 7, 8], 3, 2) == [(1, 2, 3), (3, 4, 5), (5, 6, 7), (7, 8, None)]


@pytest.mark.parametrize("length,overlap", [(10, 3), (4, 1), (3, 2), (5, 0)])
def test_window_indices_follow_windowed_semantics(length, overlap):
    from flair_amd.video import window_indices
    for n in range(0, 40):
        want = [[v for v in w if v is not None] for w in _windowed_reference(range(n), length, length - overlap)]
        want = [w for w in want if w]
        assert window_indices(n, length, overlap) == want, n
    with pytest.raises(ValueError):
        window_indices(5, 3, 3)


def _toy_model(x, t, **kw):
    """Deterministic stand-in network (6 channels out: eps | variance logits), frame-coupled through the
    conditioning so that windows matter."""
    lr = kw["low_res_input"][0]
    eps = 0.3 * x - 0.2 * lr + 0.05 * torch.roll(x, 1, 0) + 0.01 * t.view(-1, 1, 1, 1).float() / 50.0
    return torch.cat([eps, 0.1 * x], 1)


@pytest.mark.gpu
def test_restore_video_matches_oracle_loop(dev):
    """5 frames, windows of 4 with 1 frame of overlap, blur task, 6-step chain: the HIP harness against the
    same loop restated with the CPU oracle (torch resize, oracle sampler + blur operator)."""
    from flair_amd import video
    from flair_amd import workload as wl
    from flair_amd.guided_diffusion import pseudoSR as psr
    from oracle import degrade as odeg
    from oracle import diffusion as odiff
    N, s, S, L, OV, steps = 5, 8, 32, 4, 1, 6
    g = torch.Generator().manual_seed(23)
    degraded = torch.rand(1, N, 3, s, s, generator=g)
    hp = wl.TASKS["gaussian"]
    kern = wl.synthetic_blur_kernel()
    wins = video.window_indices(N, L, OV)
    tapes = [[torch.randn(len(w), 3, S, S, generator=g) for _ in range(steps)] for w in wins]
    qnoise = [torch.randn(len(w), 3, S, S, generator=g) for w in wins]

    # ---- oracle restatement of scripts/video_sample.py:371-485
    tab = odiff.Spaced(odiff.spaced_steps(1000, str(steps)), odiff.named_betas("face_blur", 1000))
    oblur = odeg.BlurOperator(kern, 4)
    prev, ref = None, []
    for wi, idx in enumerate(wins):
        d = degraded[:, idx[0]:idx[-1] + 1]
        init = F.interpolate(d[0], (S, S), mode="area").clamp(0, 1)[None]
        d_n, init_n = (d - 0.5) / 0.5, (init - 0.5) / 0.5
        T = len(idx)
        a = torch.from_numpy(tab.sqrt_alphas_cumprod).float()[tab.num_timesteps - 1]
        b = torch.from_numpy(tab.sqrt_one_minus_alphas_cumprod).float()[tab.num_timesteps - 1]
        noise = a * init_n[0] + b * qnoise[wi]
        rnn = F.interpolate(d_n[0], (S, S), mode="bicubic", align_corners=False).clamp(-1, 1)[None]
        sample = odiff.sample_loop(tab, _toy_model, noise,
                                   model_kwargs=dict(low_res_input=init_n, num_frames=T, rnn_input=rnn),
                                   restore_fn=lambda x0, _d=d_n: oblur.a_pinv(_d[0], x0),
                                   aux_model=wl.identity_aux, w=hp["w"], tau=2, rho=hp["rho"],
                                   noise_level=hp["noise_level"], zeta=hp["zeta"], prev_recon=prev,
                                   step_noise=tapes[wi])[None]
        if prev is not None:
            sample = sample[:, OV:]
        prev = sample[:, -OV:].clone()
        ref.append((sample.clamp(-1, 1) + 1) / 2)
    ref = torch.cat(ref, 1)[0]

    # ---- HIP harness
    diffusion = wl.diffusion_for(steps)
    A = psr.pseudoSR(psr.Get_pseudoSR_Conf(4), upscale_kernel=kern, kernel_indx=10).WrapArchitecture_PyTorch().to(dev)

    class M:
        def parameters(self):
            return iter([degraded.to(dev)])

        def __call__(self, x, t, **kw):
            return _toy_model(x, t, **kw)
    got = video.restore_video(
        "gaussian", degraded.to(dev), M(), diffusion, lambda d_n: (lambda x0: A.A_pinv(d_n[0].contiguous(), x0)),
        size=S, tau=2, length=L, overlap=OV, noise_fn=lambda wi, it, like: tapes[wi][it].to(dev),
        q_noise_fn=lambda wi, like: qnoise[wi].to(dev))
    torch.cuda.synchronize()
    assert got.shape == (N, 3, S, S)
    err = (got.cpu() - ref).abs().max().item()
    assert err <= 5e-4, err          # f32 elementwise chain (same bound as the sampler trajectory test)
