"""Generate the golden fixtures that pin the CPU oracle to the reference's own code.

Run ONLY in the build container (needs /root/reference):

    cd /tmp && python /root/repo/tests/golden/make_golden.py

The reference's Python executes unmodified; its absent third-party callees are the
restatements of oracle/thirdparty.py (refimport.py).  Fixtures hold inputs and expected
outputs only -- weights are regenerated from parameter NAMES (``name_seeded_weights``), so
the same values can be produced for the reference modules here and for the oracle / HIP
modules in the tests without storing hundreds of MB.
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden.weights import name_seeded_weights  # noqa: E402

SMALL = dict(image_size=32, in_channels=6, model_channels=128, out_channels=6, num_res_blocks=1,
             attention_resolutions=(2, 4), rnn_resolutions=(1, 2), channel_mult=(0.5, 1, 4), use_fp16=False,
             num_head_channels=64, resblock_updown=True, use_scale_shift_norm=True, temporal_block=True,
             use_checkpoint=False)


def save(name, **arrays):
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


def toy_model(x, t, **kw):
    tt = t.float().view(-1, 1, 1, 1) / 1000.0
    eps = torch.tanh(x * 0.7 + tt) * 0.9 + 0.1 * x.roll(1, dims=3)
    v = torch.sin(x * 1.3 - tt)
    return torch.cat([eps, v], dim=1)


class ToyModule(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x, t, **kw):
        return toy_model(x, t, **kw)


def main():
    import refimport
    refimport.install_stubs()
    gd = refimport.ref("gaussian_diffusion")
    rs = refimport.ref("respace")
    un = refimport.ref("unet_new")
    nn_new = refimport.ref("nn_new")
    torch.set_grad_enabled(False)

    # ---- G1: tables / timestep maps ---------------------------------------------------
    g1 = {}
    for sched, base, counts in (("face_blur", 1000, ("50", "100", "250")), ("face_bicubic", 2000, ("100",))):
        for c in counts:
            d = rs.SpacedDiffusion(use_timesteps=rs.space_timesteps(base, c, "uniform"),
                                   betas=gd.get_named_beta_schedule(sched, base),
                                   model_mean_type=gd.ModelMeanType.EPSILON,
                                   model_var_type=gd.ModelVarType.LEARNED_RANGE, loss_type=gd.LossType.MSE)
            key = f"{sched}_{c}"
            g1[key + "_map"] = np.array(d.timestep_map)
            for tab in ("betas", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                        "sqrt_alphas_cumprod_prev", "sqrt_one_minus_alphas_cumprod_prev",
                        "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
                g1[f"{key}_{tab}"] = getattr(d, tab)
    g1["ddim25"] = np.array(sorted(rs.space_timesteps(1000, "ddim25")))
    g1["quad20"] = np.array(rs.space_timesteps(1000, 20, "quad"))
    g1["sections"] = np.array(sorted(rs.space_timesteps(300, "10,15,20")))
    save("g1_tables", **g1)

    # ---- G3: sampler trajectories with injected noise (+ G2 ramps recorded on the way) ----
    T, S = 4, 16
    gen = torch.Generator().manual_seed(17)
    x_T = torch.randn(T, 3, S, S, generator=gen)
    prev = torch.rand(1, 2, 3, S, S, generator=gen) * 2 - 1
    tape = [torch.randn(T, 3, S, S, generator=gen) for _ in range(12)]
    restore = lambda x0: 0.3 * x0 - 0.1 * x0.flip(2)        # noqa: E731  (any deterministic operator)
    aux = lambda x0, t, xt: 0.8 * x0 + 0.1 * xt            # noqa: E731
    cases = {
        "lr_restore": dict(var=gd.ModelVarType.LEARNED_RANGE, steps="10", restore=True, prev=False, t_start=-1,
                           zeta=1.0, noise_level=2.55, w=0.75, rho=0.25, tau=2),
        "fs_prev_tstart": dict(var=gd.ModelVarType.FIXED_SMALL, steps="10", restore=False, prev=True, t_start=6,
                               zeta=-1, noise_level=None, w=0.5, rho=0.5, tau=0),
        "lr_all": dict(var=gd.ModelVarType.LEARNED_RANGE, steps="12", restore=True, prev=True, t_start=-1,
                       zeta=1.0, noise_level=12.75, w=0.5, rho=0.0, tau=5),
    }
    g3 = {"x_T": x_T, "prev": prev, "tape": torch.stack(tape)}
    for name, c in cases.items():
        d = rs.SpacedDiffusion(use_timesteps=rs.space_timesteps(1000, c["steps"], "uniform"),
                               betas=gd.get_named_beta_schedule("face_blur", 1000),
                               model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=c["var"],
                               loss_type=gd.LossType.MSE)
        it = iter(tape)
        orig = torch.randn_like
        torch.randn_like = lambda x, *a, **k: next(it)
        seen = []
        orig_ps = d.p_sample

        def spy(*a, _o=orig_ps, **k):
            seen.append((float(k["w"].reshape(-1)[0]), float(k["gamma"].reshape(-1)[0])))
            return _o(*a, **k)
        d.p_sample = spy
        try:
            outs = list(d.p_sample_loop_progressive(
                ToyModule(), x_T.shape, noise=x_T.clone(), model_kwargs=dict(num_frames=T),
                restore_fn=restore if c["restore"] else None, aux_model=aux, w=c["w"], tau=c["tau"], aligned=True,
                rho=c["rho"], noise_level=c["noise_level"], prev_recon=prev.clone() if c["prev"] else None,
                zeta=c["zeta"], t_start=c["t_start"]))
        finally:
            torch.randn_like = orig
        g3[name + "_samples"] = torch.stack([o["sample"] for o in outs])
        g3[name + "_x0"] = torch.stack([o["pred_xstart"] for o in outs])
        g3[name + "_w_gamma"] = np.array(seen)
    save("g3_sampler", **g3)

    # ---- G4/G5: network blocks and the small UNet, weights derived from parameter names ----
    with refimport.cuda_shaped():
        model = un.UNetModel(**SMALL)
    name_seeded_weights(model)
    model.eval()
    T, S = 4, 32
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(T, 3, S, S, generator=gen)
    base = torch.rand(3, S, S, generator=gen) * 2 - 1
    lr = torch.stack([torch.roll(base, shifts=(i, 2 * i), dims=(1, 2)) for i in range(T)])[None]
    lr = (lr + 0.05 * torch.randn(1, T, 3, S, S, generator=gen)).clamp(-1, 1)
    t = torch.full((T,), 371, dtype=torch.long)
    stages = {}

    def hook(nm):
        def f(mod, inp, out):
            stages[nm] = out[0] if out.dim() == 5 else out
        return f
    for i, b in enumerate(model.input_blocks):
        b.register_forward_hook(hook(f"input_blocks.{i}"))
    model.middle_block.register_forward_hook(hook("middle_block"))
    for i, b in enumerate(model.output_blocks):
        b.register_forward_hook(hook(f"output_blocks.{i}"))
    y = model(x, t, low_res_input=lr, num_frames=T, vsrpp_weights=1.0)
    ff, fb = model.compute_flow(lr)
    keep = ["input_blocks.0", "input_blocks.1", "input_blocks.2", "input_blocks.4", "middle_block",
            "output_blocks.0", "output_blocks.3", "output_blocks.5"]
    save("g5_unet_small", x=x, lr=lr, t=t, y=y, flows_forward=ff, flows_backward=fb,
         **{"stage_" + k.replace(".", "_"): stages[k].half() for k in keep})
    emb = nn_new.timestep_embedding(torch.tensor([0., 1., 37., 999., 500.5]), 128)
    save("g4_timestep_embedding", t=np.array([0., 1., 37., 999., 500.5]), emb=emb)

    # ---- G7: sr3.UNet (bicubic tasks), small configuration ---------------------------------
    sr3 = refimport.ref("sr3")
    SR3_SMALL = dict(image_size=64, in_channel=6, out_channel=3, inner_channel=64, norm_groups=16,
                     channel_mults=(1, 2, 4), attn_res=(32, 16), vsrpp_res=(64,), spatial_attn=False,
                     temporal_attn=True, res_blocks=1, dropout=0.0, dtype=torch.float32, cross_frame_module=True,
                     use_checkpoint=False, num_frames=7, head_dim=64)
    with refimport.cuda_shaped():
        m3 = sr3.UNet(**SR3_SMALL)
    name_seeded_weights(m3)
    m3.eval()
    gen = torch.Generator().manual_seed(5)
    x3 = torch.randn(4, 3, 64, 64, generator=gen)
    base3 = torch.rand(3, 64, 64, generator=gen) * 2 - 1
    lr3 = torch.stack([torch.roll(base3, shifts=(i, 2 * i), dims=(1, 2)) for i in range(4)])[None]
    lr3 = (lr3 + 0.05 * torch.randn(1, 4, 3, 64, 64, generator=gen)).clamp(-1, 1)
    lv3 = torch.full((4,), 0.83)
    y3 = m3(x3, lv3, low_res_input=lr3, num_frames=4, vsrpp_weights=0.93)
    # the _WrappedModel path for SR3: continuous noise level sqrt(acp_prev)[t+1]
    d3 = rs.SpacedDiffusion(use_timesteps=rs.space_timesteps(2000, "100", "uniform"),
                            betas=gd.get_named_beta_schedule("face_bicubic", 2000),
                            model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType.FIXED_SMALL,
                            loss_type=gd.LossType.MSE)
    seen = {}

    class Probe(sr3.UNet):
        def __init__(self):
            torch.nn.Module.__init__(self)

        def forward(self, x, level, **kw):
            seen["level"] = level
            return torch.zeros_like(x)
    d3._wrap_model(Probe())(x3, torch.tensor([0, 17, 50, 99]))
    save("g7_sr3_small", x=x3, lr=lr3, level=lv3, y=y3, wrapped_levels=seen["level"])

    # ---- G6: degradation operators ------------------------------------------------------
    import scipy.io
    jp = refimport.ref("jpeg")
    ps = refimport.ref("pseudoSR")
    ru = refimport.ref("restore_util")
    rz = refimport.ref("resizer")
    K = scipy.io.loadmat(os.path.join(refimport.REFERENCE_ROOT, "miscs", "kernels_12.mat"))["kernels"][0, 3]
    conf = ps.Get_pseudoSR_Conf(4)
    conf.sigmoid_range_limit = False
    op = ps.pseudoSR(conf, upscale_kernel=K, kernel_indx=10)
    A = op.WrapArchitecture_PyTorch()
    gen = torch.Generator().manual_seed(11)
    img = torch.rand(2, 3, 64, 64, generator=gen) * 2 - 1
    low = torch.rand(2, 3, 16, 16, generator=gen) * 2 - 1
    g6 = {"kernel_0_3": K, "ds_kernel": op.ds_kernel, "inv_hTh": op.inv_hTh, "img": img, "low": low,
          "a_pinv": A.A_pinv(low, img), "a_pinv_lr_only": A.A_pinv(low)}
    for qf in (10, 60, 90):
        enc = jp.jpeg_encode(img.clone(), qf)
        g6[f"jpeg{qf}_luma"], g6[f"jpeg{qf}_chroma"] = enc[0], enc[1]
        g6[f"jpeg{qf}_dec"] = jp.jpeg_decode([e.clone() for e in enc], qf)
    g6["a_pinv_jpeg60"] = A.A_pinv(low, img, jpeg_encode=lambda v: jp.jpeg_encode(v, 60),
                                   jpeg_decode=lambda v: jp.jpeg_decode(v, 60))
    for f in (8, 16):
        k = np.zeros(f * 4)
        for i in range(f * 4):
            xx = abs((1 / f) * (i - np.floor(f * 4 / 2) + 0.5))
            a = -0.5
            k[i] = ((a + 2) * xx ** 3 - (a + 3) * xx ** 2 + 1) if xx <= 1 else (
                (a * xx ** 3 - 5 * a * xx ** 2 + 8 * a * xx - 4 * a) if xx < 2 else 0)
        k = torch.from_numpy(k / k.sum()).float()
        sr = ru.SRConv(k / k.sum(), 3, 64, torch.device("cpu"), stride=f)
        v = img.reshape(2, -1)
        yy = sr.A(v)
        g6[f"srconv{f}_A"] = yy
        g6[f"srconv{f}_pinv"] = sr.A_pinv(yy)
    g6["resizer_down8"] = rz.Resizer(img.shape, 1 / 8)(img)
    g6["resizer_up8"] = rz.Resizer(low.shape, 8)(low)
    save("g6_degrade", **g6)


if __name__ == "__main__":
    main()
