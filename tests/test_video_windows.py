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


def _write_pngs(tmp_path, n, s, seed=5):
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    names = [f"frame_{i}.png" for i in range(n)]           # natural order differs from lexicographic order (10 < 2)
    frames = []
    for nm in names:
        arr = rng.integers(0, 256, size=(s, s, 3), dtype=np.uint8)
        Image.fromarray(arr, mode="RGB").save(tmp_path / nm)
        frames.append(arr)
    (tmp_path / "notes.txt").write_text("not a frame")
    return names, frames


def test_frame_files_round_trip(tmp_path):
    """flair_amd.io: the reference's glob + natural order (video_sample.py:334), decode to (1,N,3,h,w) float/255
    (:337-345), and ``(x*255).byte()`` -> PNG (:487-492) give back the same bytes."""
    import numpy as np
    from PIL import Image
    from flair_amd import io as fio
    names, frames = _write_pngs(tmp_path, 12, 8)
    paths = fio.list_frames(tmp_path)
    assert [p.split("/")[-1] for p in paths] == names      # frame_2 before frame_10, notes.txt excluded
    x = fio.read_frames(paths)
    assert x.shape == (1, 12, 3, 8, 8) and x.dtype == torch.float32
    assert torch.equal((x[0] * 255).round().byte(), torch.from_numpy(np.stack(frames)).permute(0, 3, 1, 2))
    u8 = fio.to_bytes(x[0])
    assert u8.shape == (12, 8, 8, 3)
    w = fio._Writer(tmp_path / "out")
    w.submit(0, x[0][:7])
    w.submit(7, x[0][7:])
    w.close()
    back = [np.asarray(Image.open(tmp_path / "out" / f"{i:04d}.png")) for i in range(12)]
    # float/255*255 truncates to v or v-1 (the reference's own behaviour: it never rounds, :489)
    assert all(np.abs(b.astype(int) - f.astype(int)).max() <= 1 for b, f in zip(back, frames))
    assert all(np.array_equal(b, np.asarray(u8[i])) for i, b in enumerate(back))


def test_iter_windows_on_host(tmp_path):
    """The streaming window iterator (decode-ahead thread) yields the windows of window_indices with the
    same frames read_frames gives, each shared frame decoded once."""
    from flair_amd import io as fio
    from flair_amd.video import window_indices
    _write_pngs(tmp_path, 9, 6)
    paths = fio.list_frames(tmp_path)
    full = fio.read_frames(paths)
    got = list(fio.iter_windows(paths, "cpu", length=4, overlap=1))
    assert [idx for idx, _ in got] == window_indices(9, 4, 1)
    for idx, x in got:
        assert torch.equal(x, full[:, idx[0]:idx[-1] + 1])


@pytest.mark.gpu
def test_restore_video_files_matches_in_memory(dev, tmp_path):
    """File-to-file streaming harness == the in-memory loop on the same frames (toy network, 4-step chain)."""
    import numpy as np
    from PIL import Image
    from flair_amd import io as fio
    from flair_amd import video
    from flair_amd import workload as wl
    from flair_amd.guided_diffusion import pseudoSR as psr
    N, s, S, L, OV, steps = 7, 8, 32, 4, 1, 4
    _write_pngs(tmp_path, N, s)
    paths = fio.list_frames(tmp_path)
    degraded = fio.read_frames(paths).to(dev)
    g = torch.Generator().manual_seed(3)
    wins = video.window_indices(N, L, OV)
    tapes = [[torch.randn(len(w), 3, S, S, generator=g).to(dev) for _ in range(steps)] for w in wins]
    qnoise = [torch.randn(len(w), 3, S, S, generator=g).to(dev) for w in wins]
    diffusion = wl.diffusion_for(steps)
    A = psr.pseudoSR(psr.Get_pseudoSR_Conf(4), upscale_kernel=wl.synthetic_blur_kernel(),
                     kernel_indx=10).WrapArchitecture_PyTorch().to(dev)

    class M:
        def parameters(self):
            return iter([degraded])

        def __call__(self, x, t, **kw):
            return _toy_model(x, t, **kw)
    common = dict(size=S, tau=1, length=L, overlap=OV, noise_fn=lambda wi, it, like: tapes[wi][it],
                  q_noise_fn=lambda wi, like: qnoise[wi])
    rf = lambda d_n: (lambda x0: A.A_pinv(d_n[0].contiguous(), x0))      # noqa: E731
    ref = video.restore_video("gaussian", degraded, M(), diffusion, rf, **common)
    n = fio.restore_video_files("gaussian", tmp_path, tmp_path / "out", M(), diffusion, rf, device=dev, **common)
    assert n == N
    want = fio.to_bytes(ref).cpu().numpy()
    for i in range(N):
        assert np.array_equal(np.asarray(Image.open(tmp_path / "out" / f"{i:04d}.png")), want[i])


@pytest.mark.gpu
def test_restore_video_files_with_hip_graph_over_windows(dev, tmp_path):
    """The streaming harness with a REAL (reduced-width) UNetModel replayed from hipGraphs: 4 windows, so the
    reader thread pins / uploads window w+1 and the writer thread waits on events while the main thread warms up
    and captures the next window's graph (thread-local capture mode).  Same frames, same noise tapes: the
    graph run must equal the eager run bit for bit, file by file."""
    import numpy as np
    from PIL import Image
    from flair_amd import io as fio
    from flair_amd import video
    from flair_amd import workload as wl
    from flair_amd.guided_diffusion import pseudoSR as psr
    from flair_amd.guided_diffusion.unet_new import UNetModel
    from tests.test_gpu_unet import SMALL
    N, s, S, L, OV, steps = 10, 8, 32, 4, 1, 3
    _write_pngs(tmp_path, N, s)
    torch.manual_seed(0)
    m = UNetModel(**SMALL)
    wl.randomize_zero_modules(m)
    m = m.to(dev).eval()
    m.convert_to_fp16()
    g = torch.Generator().manual_seed(3)
    wins = video.window_indices(N, L, OV)
    assert len(wins) >= 3
    tapes = [[torch.randn(len(w), 3, S, S, generator=g).to(dev) for _ in range(steps)] for w in wins]
    qnoise = [torch.randn(len(w), 3, S, S, generator=g).to(dev) for w in wins]
    diffusion = wl.diffusion_for(steps)
    A = psr.pseudoSR(psr.Get_pseudoSR_Conf(4), upscale_kernel=wl.synthetic_blur_kernel(),
                     kernel_indx=10).WrapArchitecture_PyTorch().to(dev)
    common = dict(size=S, tau=1, length=L, overlap=OV, noise_fn=lambda wi, it, like: tapes[wi][it],
                  q_noise_fn=lambda wi, like: qnoise[wi])
    rf = lambda d_n: (lambda x0: A.A_pinv(d_n[0].contiguous(), x0))      # noqa: E731
    n = fio.restore_video_files("gaussian", tmp_path, tmp_path / "eager", m, diffusion, rf, device=dev, **common)
    assert n == N
    m.enable_hip_graph()
    n = fio.restore_video_files("gaussian", tmp_path, tmp_path / "graph", m, diffusion, rf, device=dev, **common)
    m.enable_hip_graph(False)
    assert n == N
    for i in range(N):
        a = np.asarray(Image.open(tmp_path / "eager" / f"{i:04d}.png"))
        b = np.asarray(Image.open(tmp_path / "graph" / f"{i:04d}.png"))
        assert np.array_equal(a, b), i


@pytest.mark.gpu
def test_window_with_hip_prior_and_parsing_weights(dev):
    """scripts/video_sample.py:427-479 for one x16-bicubic window at the reference's 512x512: per-pixel
    ``vsrpp_weights`` from ParseNet's class-0 mask and the CodeFormer prior blended into every step, both on the HIP
    kernels (workload.parsenet_weights_fn / codeformer_aux), against the same window driven by the CPU oracles of the
    two networks (2 frames, 2-step chain, toy eps-network; the oracle's code indices are injected so that an arg-max
    near-tie cannot fork the runs)."""
    from flair_amd import video
    from flair_amd import workload as wl
    from flair_amd.guided_diffusion.codeformer import CodeFormer
    from flair_amd.guided_diffusion.parsenet import ParseNet
    from oracle import codeformer as ocf
    from oracle import parsenet as opn
    from tests.golden.make_golden import codeformer_input
    from tests.golden.weights import name_seeded_weights
    T, S, steps = 2, 512, 2
    gan = name_seeded_weights(CodeFormer()).eval()
    parser = name_seeded_weights(ParseNet(in_size=512, out_size=512, parsing_ch=19)).eval()
    gan_sd = {k: v.detach().clone() for k, v in gan.state_dict().items()}
    logits_probe = opn.parsenet_forward({k: v.detach().clone() for k, v in parser.state_dict().items()},
                                        codeformer_input(batch=1, seed=5))[0]
    with torch.no_grad():      # make the background class cover about half of the frame (see tests/test_parsenet.py)
        parser.out_mask_conv.conv2d.bias[0] += float((logits_probe[:, 1:].max(1)[0] - logits_probe[:, 0]).median())
    parser_sd = {k: v.detach().clone() for k, v in parser.state_dict().items()}
    gan, parser = gan.to(dev), parser.to(dev)
    g = torch.Generator().manual_seed(77)
    degraded = ((codeformer_input(batch=T, seed=5)[..., ::16, ::16] + 1) / 2).clamp(0, 1)[None]      # (1, T, 3, 32, 32)
    tape = [torch.randn(T, 3, S, S, generator=g) for _ in range(steps)]
    qn = torch.randn(T, 3, S, S, generator=g)
    diffusion = wl.bicubic_diffusion_for(steps)
    seen = {}

    class M:
        def parameters(self):
            return iter([degraded.to(dev)])

        def __call__(self, x, t, **kw):
            seen.setdefault("weights", []).append(kw["vsrpp_weights"])
            lr = kw["low_res_input"][0]
            return 0.3 * x - 0.2 * lr + 0.05 * torch.roll(x, 1, 0)

    def run(aux_model, weights_fn):
        seen.clear()
        out, _ = video.restore_window("x16_bicubic", degraded.to(dev), M(), diffusion, lambda d_n: None, size=S,
                                      aux_model=aux_model, vsrpp_weights_fn=weights_fn, tau=0,
                                      noise_fn=lambda wi, it, like: tape[it].to(dev), q_noise_fn=lambda wi, like: qn.to(dev))
        torch.cuda.synchronize()
        return out.cpu(), seen["weights"][0].cpu()

    codes = []

    def oracle_aux(x0, *a, **k):
        out, logits, _ = ocf.codeformer_forward(gan_sd, x0.cpu(), w=1.0, adain=True)
        codes.append(logits.argmax(2))
        return out.to(dev)

    def oracle_weights(init_norm):
        return opn.face_weight(parser_sd, init_norm[0].cpu(), wl.TASKS["x16_bicubic"]["face_weight"])[None].to(dev)

    ref, ref_w = run(oracle_aux, oracle_weights)
    assert len(codes) == steps
    it = iter(codes)
    got, got_w = run(lambda x0, *a, **k: gan(x0, w=1.0, adain=True, code_idx=next(it))[0],
                     wl.parsenet_weights_fn(parser, "x16_bicubic"))
    assert got.shape == ref.shape == (T, 3, S, S) and got_w.shape == ref_w.shape == (1, T, 1, S, S)
    assert (got_w != ref_w).float().mean().item() < 1e-3          # parsing map: equal up to arg-max near-ties
    assert 0.02 < (ref_w < 1).float().mean().item() < 0.98
    assert (got - ref).abs().max().item() <= 1e-3
