"""Timestep respacing (reference: guided_diffusion/respace.py) for the MI355X sampler."""
import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts, mode="uniform"):
    """Which original timesteps a shortened chain keeps (respace.py:7-66).

    "uniform": per section, ``count`` steps at a fractional stride (``"ddimN"`` = integer
    stride giving exactly N steps); "quad": quadratically spaced list.
    """
    if mode == "quad":
        seq = np.linspace(0, np.sqrt(num_timesteps * 0.8), int(section_counts)) ** 2
        return [int(s) for s in list(seq)]
    if mode != "uniform":
        return None
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(num_timesteps, len(section_counts))
    taken, first = [], 0
    for k, count in enumerate(section_counts):
        size = per + (1 if k < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            taken.append(first + round(pos))
            pos += stride
        first += size
    return set(taken)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process that keeps a subset of the base steps (respace.py:78-135)."""

    def __init__(self, use_timesteps, noise_schedule="linear", **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.noise_schedule = noise_schedule
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        last, new_betas = 1.0, []
        for i, acp in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - acp / last)
                last = acp
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        # one wrapper per (diffusion, model): its timestep table is uploaded once, not once per denoising step (the
        # upload is a host-device synchronisation in the middle of every step: tools/probes/sync_probe.py)
        cache = self.__dict__.setdefault("_wrapped", {})
        w = cache.get(id(model))
        if w is None or w.model is not model:
            if len(cache) >= 8:
                cache.clear()
            w = _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps,
                              noise_schedule=self.noise_schedule,
                              sqrt_alphas_cumprod_prev=self.sqrt_alphas_cumprod_prev)
            cache[id(model)] = w
        return w

    def _scale_timesteps(self, t):
        return t  # done by the wrapped model


class _WrappedModel:
    """Maps loop indices to original timesteps before calling the network (respace.py:138-167).
    The map lives on the GPU once; SR3-style models (attribute ``takes_noise_level``) receive
    the continuous level sqrt(acp_prev)[t+1] instead."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps,
                 noise_schedule="linear", sqrt_alphas_cumprod_prev=None):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps
        self.noise_schedule = noise_schedule
        self.sqrt_alphas_cumprod_prev = sqrt_alphas_cumprod_prev
        self._map = {}

    def _table(self, key, values, device, dtype):
        k = (key, device, dtype)
        if k not in self._map:
            self._map[k] = th.tensor(values, device=device, dtype=dtype)
        return self._map[k]

    def parameters(self):
        return self.model.parameters()

    def __call__(self, x, ts, **kwargs):
        kwargs["old_ts"] = ts
        if getattr(self.model, "takes_noise_level", False):
            levels = self._table("lvl", np.asarray(self.sqrt_alphas_cumprod_prev, dtype=np.float32),
                                 x.device, th.float32)
            return self.model(x, levels[ts + 1], **kwargs)
        new_ts = self._table("map", self.timestep_map, ts.device, ts.dtype)[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
