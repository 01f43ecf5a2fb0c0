"""CPU restatement of FLAIR's sampler (numpy fp64 tables -> torch fp32 tensors).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
guided_diffusion/gaussian_diffusion.py and guided_diffusion/respace.py of the
reference (line numbers cited per function).  Only the branches reachable from
scripts/video_sample.py are restated: EPSILON mean type, LEARNED_RANGE or
FIXED_SMALL variance, ``sample_mode="ddpm"`` (which in FLAIR is a generalised DDIM
step with stochasticity ``rho``, gaussian_diffusion.py:507-515).
"""
import numpy as np
import torch


def named_betas(name, n):
    """gaussian_diffusion.py:15-36."""
    if name == "face_blur":
        s = 1000 / n
        return np.linspace(s * 1e-4, s * 2e-2, n, dtype=np.float64)
    if name == "face_bicubic":
        return np.linspace(1e-6, 1e-2, 2000, dtype=np.float64)
    raise NotImplementedError(f"unknown beta schedule: {name}")


def spaced_steps(num_timesteps, section_counts, mode="uniform"):
    """respace.py:7-66 -- which original timesteps a shortened chain keeps."""
    if mode == "quad":
        seq = np.linspace(0, np.sqrt(num_timesteps * 0.8), int(section_counts)) ** 2
        return [int(s) for s in seq]
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(v) for v in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    kept, start = [], 0
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class Tables:
    """Coefficient tables of gaussian_diffusion.py:112-173 for a given beta vector."""

    def __init__(self, betas):
        b = np.asarray(betas, dtype=np.float64)
        assert b.ndim == 1 and (b > 0).all() and (b <= 1).all()
        self.betas = b
        self.num_timesteps = len(b)
        a = 1.0 - b
        ac = np.cumprod(a)
        ac_prev = np.append(1.0, ac[:-1])
        self.alphas_cumprod = ac
        self.alphas_cumprod_prev = ac_prev
        self.sqrt_alphas_cumprod_prev = np.sqrt(np.append(1.0, ac))          # length T+1
        self.sqrt_alphas_cumprod = np.sqrt(ac)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac)
        self.sqrt_one_minus_alphas_cumprod_prev = np.append(0.0, np.sqrt(1.0 - ac[:-1]))
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        pv = b * (1.0 - ac_prev) / (1.0 - ac)
        self.posterior_variance = pv
        self.posterior_log_variance_clipped = np.log(np.append(pv[1], pv[1:]))
        self.posterior_mean_coef1 = b * np.sqrt(ac_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - ac_prev) * np.sqrt(a) / (1.0 - ac)


class Spaced(Tables):
    """respace.py:78-102: betas re-derived from the kept alphas_cumprod + timestep_map."""

    def __init__(self, use_timesteps, betas):
        base = Tables(betas)
        use = set(use_timesteps)
        self.timestep_map, new_betas, last = [], [], 1.0
        for i, ac in enumerate(base.alphas_cumprod):
            if i in use:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        self.original_num_steps = len(betas)
        super().__init__(np.array(new_betas))


def _at(arr, t, like):
    """gaussian_diffusion.py:692-705."""
    v = torch.from_numpy(np.asarray(arr))[t].float()
    return v.reshape(-1, *([1] * (like.dim() - 1))).expand(like.shape)


def q_sample(tab, x0, t, noise):
    """gaussian_diffusion.py:206-224."""
    return _at(tab.sqrt_alphas_cumprod, t, x0) * x0 + _at(tab.sqrt_one_minus_alphas_cumprod, t, x0) * noise


def aux_weights(tab, start, tau, w, have_aux=True):
    """The ``ws`` ramp of gaussian_diffusion.py:632-646."""
    T = tab.num_timesteps
    if not have_aux:
        return np.ones(T)
    if start - tau > 0:
        ws = np.exp(-np.linspace(0, 1, start - tau + 1))
        ws = 1 - (ws - ws.min()) / (ws.max() - ws.min()) * (1 - w)
        ws = np.append(ws, np.ones(T - start - 1))
        return np.concatenate([np.ones(tau), ws])
    return np.ones(T) * w


def consistency_gammas(tab, zeta, noise_level):
    """gaussian_diffusion.py:648-657."""
    if zeta == -1:
        return np.ones_like(tab.betas)
    g = zeta * (noise_level ** 2 / (tab.sqrt_one_minus_alphas_cumprod / tab.sqrt_alphas_cumprod) ** 2)
    g[g >= 1] = 0.991
    g[g <= 1e-1] = 1e-6
    return 1 - g


def sample_loop(tab, model, x_T, *, model_kwargs, learned_range=True, restore_fn=None,
                aux_model=None, w=0.5, tau=0, rho=0.35, noise_level=None, zeta=-1,
                prev_recon=None, t_start=-1, step_noise=None, clip_denoised=True,
                sr3_noise_level=False, trace=None, aligned=True, face_restore_helper=None, affine_matrices=None):
    """p_sample_loop_progressive + p_sample + p_mean_variance (gaussian_diffusion.py:250-342, 423-517, 589-689; model
    wrapping respace.py:155-167).  ``aligned=False`` takes the crop / paste branch of :476-493 through
    ``face_restore_helper`` (an object with the reference helper's ``get_crop_face_from_affine_matrices`` and
    ``inverse_faces``; tests pass one built on oracle/facewarp.py).

    ``step_noise``: list of per-iteration gaussian tensors replacing ``randn_like``
    (index 0 = first executed step), so CPU and GPU runs can share the draw.
    ``trace``: optional list receiving (t, pred_xstart, sample) per step.
    """
    T = tab.num_timesteps
    idx = list(range(T))
    if t_start != -1:
        if t_start < 0 or t_start >= T:
            raise ValueError("t_start must be in [0, num_timesteps)")
        idx = idx[: t_start + 1]
    idx = idx[::-1]
    ws = aux_weights(tab, idx[0], tau, w, aux_model is not None)
    gammas = consistency_gammas(tab, zeta, noise_level)
    tmap = torch.tensor(getattr(tab, "timestep_map", list(range(T))))
    img = x_T
    n = img.shape[0]
    for it, i in enumerate(idx):
        t = torch.full((n,), i, dtype=torch.long)
        if sr3_noise_level:
            model_t = torch.from_numpy(tab.sqrt_alphas_cumprod_prev).float()[t + 1]
        else:
            model_t = tmap[t]
        out = model(img, model_t, **model_kwargs)
        c = img.shape[1]
        if learned_range:
            eps, _var = torch.split(out, c, dim=1)
        else:
            eps = out[:, :3] if out.shape[1] == 6 else out
        x0 = _at(tab.sqrt_recip_alphas_cumprod, t, img) * img - _at(tab.sqrt_recipm1_alphas_cumprod, t, img) * eps
        if clip_denoised:
            x0 = x0.clamp(-1, 1)
        if restore_fn is not None:
            x0 = x0 - _at(gammas, t, img) * restore_fn(x0)
            if clip_denoised:
                x0 = x0.clamp(-1, 1)
        if aux_model is not None and i <= idx[0] and i >= tau:
            if aligned:
                face = aux_model(x0, t, img)
            else:
                aux_face = face_restore_helper.get_crop_face_from_affine_matrices(x0, affine_matrices)
                aux_xt = face_restore_helper.get_crop_face_from_affine_matrices(img, affine_matrices)
                aux_face = aux_model(aux_face, t, aux_xt)
                inv_face, inv_mask = face_restore_helper.inverse_faces(aux_face, affine_matrices)
                face = x0 * (1 - inv_mask) + inv_face * inv_mask
            if clip_denoised:
                face = face.clamp(-1, 1)
            wt = _at(ws, t, img)
            x0 = wt * x0 + (1 - wt) * face
        if prev_recon is not None:
            nf = model_kwargs["num_frames"]
            x0 = x0.reshape(-1, nf, *x0.shape[1:]).clone()
            x0[:, : prev_recon.shape[1]] = prev_recon
            x0 = x0.reshape(-1, *x0.shape[2:])
        eps2 = (_at(tab.sqrt_recip_alphas_cumprod, t, img) * img - x0) / _at(tab.sqrt_recipm1_alphas_cumprod, t, img)
        z = step_noise[it] if step_noise is not None else torch.randn_like(img)
        co = _at(tab.sqrt_one_minus_alphas_cumprod_prev, t, img)
        nz = (t != 0).float().reshape(-1, *([1] * (img.dim() - 1)))
        img = _at(tab.sqrt_alphas_cumprod_prev, t, img) * x0 + nz * (
            np.sqrt(1 - rho) * co * eps2 + np.sqrt(rho) * co * z)
        if trace is not None:
            trace.append((i, x0, img))
    return img
