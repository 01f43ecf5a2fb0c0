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


def h16(t):
    """Round through fp16 so a fixture input can be stored in half the bytes and still be exact."""
    return t.half().float()


def g4_blocks():
    """G4: per-block inputs / outputs of the reference's own block classes, weights from
    parameter names.  Inputs are stored as fp16 (the values are fp16-exact), outputs as f32."""
    import refimport
    un = refimport.ref("unet_new")
    sr3 = refimport.ref("sr3")
    unet_old = refimport.ref("unet_old")
    out = {}

    def seeded(name):
        return torch.Generator().manual_seed(zlib.crc32(name.encode()))

    def run(name, block, shape, emb_dim=512, with_emb=True, call=None):
        g = seeded(name)
        x = h16(torch.randn(1, *shape, generator=g))
        emb = h16(torch.randn(shape[0], emb_dim, generator=g)) if with_emb else None
        name_seeded_weights(block)
        block.eval()
        y = call(block, x, emb) if call else (block(x, emb) if with_emb else block(x))
        out[name + "_x"] = x.half()
        if emb is not None:
            out[name + "_emb"] = emb.half()
        out[name + "_y"] = y.float()

    R = lambda cin, cout, **kw: un.ResBlock(cin, 512, 0.0, out_channels=cout, use_scale_shift_norm=True, **kw)  # noqa: E731
    run("res2d_same", R(64, 64), (2, 64, 32, 32))
    run("res2d_skip", R(128, 64), (2, 128, 16, 16))
    run("res2d_up", R(64, 64, up=True), (2, 64, 16, 16))
    run("res2d_down", R(64, 64, down=True), (2, 64, 32, 32))
    run("res3d", un.TemporalWrapper(R(64, 64, dims=3)), (4, 64, 16, 16))
    run("attn_legacy", un.AttentionBlock(128, num_head_channels=64), (2, 128, 16, 16), with_emb=False)
    run("attn_new", un.AttentionBlock(128, num_head_channels=64, use_new_attention_order=True), (2, 128, 16, 16),
        with_emb=False)
    run("attn_bottle", un.AttentionbottleBlock(512, num_head_channels=64), (2, 512, 4, 4))
    run("tattn", un.TemporalWrapper(un.TemporalAttention(128, 5, num_head_channels=64)), (6, 128, 8, 8),
        with_emb=False)
    # raw attention functions from the stub-free unet_old (qkv: (N, 3*H*C, L))
    g = seeded("qkv")
    qkv = h16(torch.randn(3, 3 * 2 * 64, 64, generator=g))
    out["qkv_x"] = qkv.half()
    out["qkv_legacy_y"] = unet_old.QKVAttentionLegacy(2)(qkv)
    out["qkv_new_y"] = unet_old.QKVAttention(2)(qkv)

    # deformable alignment and one BasicVSR++ instance (c = 64, 32x32: the halo-conv geometry)
    def smooth_flow(g, n, h, w, mag):
        f = torch.randn(n, 2, h // 8, w // 8, generator=g) * mag
        return h16(torch.nn.functional.interpolate(f, size=(h, w), mode="bilinear", align_corners=False))

    with refimport.cuda_shaped():
        vs = un.BasicVSRPP(mid_channels=64)
    name_seeded_weights(vs)
    vs.eval()
    al = vs.deform_align["backward_1"]
    g = seeded("align")
    c, S = 64, 32
    x = h16(torch.randn(1, 2 * c, S, S, generator=g))
    extra = h16(torch.randn(1, 3 * c, S, S, generator=g))
    f1, f2 = smooth_flow(g, 1, S, S, 2.0), smooth_flow(g, 1, S, S, 3.0)
    out.update(align_x=x.half(), align_extra=extra.half(), align_flow1=f1.half(), align_flow2=f2.half(),
               align_y=al(x, extra, f1, f2).float())
    g = seeded("vsrpp")
    T = 4
    hid = h16(torch.randn(1, T, c, S, S, generator=g))
    ff = smooth_flow(g, T - 1, S, S, 1.5)[None]
    fb = smooth_flow(g, T - 1, S, S, 1.5)[None]
    out.update(vsrpp_x=hid.half(), vsrpp_ff=ff.half(), vsrpp_fb=fb.half(),
               vsrpp_y=vs(hid, ff, fb, 0.93).float())
    wmap = h16(torch.rand(1, T, 1, 16, 16, generator=g))
    out.update(vsrpp_wmap=wmap.half(), vsrpp_wmap_y=vs(hid, ff, fb, wmap).float())

    # sr3 block: ResnetBlock + (3,1,1) temporal ResBlock + 7-frame temporal attention, both emb-gated
    blk = sr3.ResnetBlocWithAttn(64, 128, noise_level_emb_dim=64, norm_groups=16, conv_3d=True, temporal_attn=True,
                                 num_frames=7, head_dim=64)
    run("sr3_block", blk, (5, 64, 16, 16), emb_dim=64, call=lambda b, x, e: b(x, None, e))
    save("g4_blocks", **out)


def g8_blur_forward():
    """pseudoSR forward operator A (pseudoSR.py:283-295, reflect padding) on a seeded image."""
    import scipy.io
    import refimport
    ps = refimport.ref("pseudoSR")
    K = scipy.io.loadmat(os.path.join(refimport.REFERENCE_ROOT, "miscs", "kernels_12.mat"))["kernels"][0, 3]
    A = ps.pseudoSR(ps.Get_pseudoSR_Conf(4), upscale_kernel=K, kernel_indx=10).WrapArchitecture_PyTorch()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(8))
    save("g8_blur_forward", x=x, y=A.A(x))


def g5_unet_64():
    """G5 at 64x64x4: the SMALL network at image_size 64 -- final output and three stages."""
    import refimport
    un = refimport.ref("unet_new")
    cfg = dict(SMALL, image_size=64)
    with refimport.cuda_shaped():
        model = un.UNetModel(**cfg)
    name_seeded_weights(model)
    model.eval()
    T, S = 4, 64
    gen = torch.Generator().manual_seed(64)
    x = h16(torch.randn(T, 3, S, S, generator=gen))
    base = torch.rand(3, S, S, generator=gen) * 2 - 1
    lr = torch.stack([torch.roll(base, shifts=(i, 2 * i), dims=(1, 2)) for i in range(T)])[None]
    lr = h16((lr + 0.05 * torch.randn(1, T, 3, S, S, generator=gen)).clamp(-1, 1))
    t = torch.full((T,), 123, dtype=torch.long)
    stages = {}
    for nm, b in (("input_blocks.2", model.input_blocks[2]), ("middle_block", model.middle_block),
                  ("output_blocks.4", model.output_blocks[4])):
        b.register_forward_hook(lambda mod, inp, o, nm=nm: stages.__setitem__(nm, o[0]))
    y = model(x, t, low_res_input=lr, num_frames=T, vsrpp_weights=1.0)
    save("g5_unet_64", x=x.half(), lr=lr.half(), t=t, y=y,
         **{"stage_" + k.replace(".", "_"): v.half() for k, v in stages.items()})


def codeformer_input(batch=1, seed=9):
    """A smooth face-sized field in [-1, 1] plus pixel noise (shared with tests/test_codeformer.py)."""
    gen = torch.Generator().manual_seed(seed)
    low = torch.randn(batch, 3, 32, 32, generator=gen)
    x = torch.nn.functional.interpolate(low, size=(512, 512), mode="bicubic", align_corners=False) * 0.6
    return h16((x + 0.05 * torch.randn(batch, 3, 512, 512, generator=gen)).clamp(-1, 1))


def g9_codeformer():
    """G9: the reference's CodeFormer, built as scripts/video_sample.py:351-357 builds it,
    name-seeded weights, called as the sampler's aux_model calls it (``gan(x0, w=1.0, adain=True)``,
    video_sample.py:450-452) plus the w = 0 / no-AdaIN variant on a pixel subset."""
    import refimport
    cf = refimport.ref("codeformer")
    gan = cf.CodeFormer(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9, connect_list=["32", "64", "128", "256"])
    name_seeded_weights(gan)
    gan.eval()
    x = codeformer_input()
    out, logits, lq = gan(x, w=1.0, adain=True)
    out0 = gan(x, w=0, adain=False)[0]
    sd = gan.state_dict()
    save("g9_codeformer", x=x.half(), out=out, logits=logits, lq_feat=lq, out_w0_sub=out0[..., ::4, ::4],
         param_names=np.array(list(sd.keys())), param_shapes=np.array([";".join(map(str, v.shape)) for v in sd.values()]))


def g10_parsenet():
    """G10 (parsing half): the reference's ParseNet as facelib/parsing/__init__.py:13-14 builds it,
    name-seeded weights and BatchNorm statistics, eval mode.  The file only imports numpy / torch, so it is loaded
    by path (the facelib package __init__ needs cv2)."""
    import importlib.util
    import refimport
    path = os.path.join(refimport.REFERENCE_ROOT, "guided_diffusion", "facelib", "parsing", "parsenet.py")
    spec = importlib.util.spec_from_file_location("ref_parsenet", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    net = mod.ParseNet(in_size=512, out_size=512, parsing_ch=19)
    name_seeded_weights(net)
    net.eval()
    x = codeformer_input(batch=1, seed=10)
    out_mask, out_img = net(x)
    sd = net.state_dict()
    save("g10_parsenet", x=x.half(), mask_sub=out_mask[..., ::4, ::4], img_sub=out_img[..., ::4, ::4],
         argmax=out_mask.argmax(1).to(torch.uint8),
         margin_sub=(out_mask.topk(2, dim=1)[0][:, 0] - out_mask.topk(2, dim=1)[0][:, 1])[..., ::4, ::4],
         param_names=np.array(list(sd.keys())), param_shapes=np.array([";".join(map(str, v.shape)) for v in sd.values()]))


def _load_ref_file(modname, *relpath):
    import importlib.util
    import refimport
    path = os.path.join(refimport.REFERENCE_ROOT, "guided_diffusion", *relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def retinaface_feats(seed=11, hw=(128, 160)):
    """Stand-ins for the ResNet-50 body's layer2 / layer3 / layer4 outputs of a 128 x 160 image (strides 8 / 16 / 32),
    fp16-exact so that the fixture can store them as halves."""
    g = torch.Generator().manual_seed(seed)
    H, W = hw
    return [torch.randn(1, c, -(-H // s), -(-W // s), generator=g).half().float() for c, s in ((512, 8), (1024, 16), (2048, 32))]


def g11_retinaface():
    """G11 (detection half): the reference's own FPN / SSH / heads (retinaface_net.py, imports torch only) wired
    as RetinaFace.__init__ / forward wire them (retinaface.py:104-156, cfg_re50: in_channel 256, out_channel 256, anchors 2),
    name-seeded weights and BatchNorm statistics under the names the full model gives them, eval mode; plus PriorBox / decode /
    decode_landm of retinaface_utils.py on the result.  (retinaface.py itself needs cv2 and torchvision's resnet50: the body is
    not part of this fixture.)"""
    import torch.nn as nn
    import torch.nn.functional as F
    net = _load_ref_file("ref_retinaface_net", "facelib", "detection", "retinaface", "retinaface_net.py")
    utl = _load_ref_file("ref_retinaface_utils", "facelib", "detection", "retinaface", "retinaface_utils.py")

    class Neck(nn.Module):
        def __init__(self):
            super().__init__()
            self.fpn = net.FPN([512, 1024, 2048], 256)
            self.ssh1, self.ssh2, self.ssh3 = net.SSH(256, 256), net.SSH(256, 256), net.SSH(256, 256)
            self.ClassHead = net.make_class_head(fpn_num=3, inchannels=256)
            self.BboxHead = net.make_bbox_head(fpn_num=3, inchannels=256)
            self.LandmarkHead = net.make_landmark_head(fpn_num=3, inchannels=256)

        def forward(self, out):
            fpn = self.fpn(out)
            features = [self.ssh1(fpn[0]), self.ssh2(fpn[1]), self.ssh3(fpn[2])]
            bbox = torch.cat([self.BboxHead[i](f) for i, f in enumerate(features)], dim=1)
            cls = torch.cat([self.ClassHead[i](f) for i, f in enumerate(features)], dim=1)
            ldm = torch.cat([self.LandmarkHead[i](f) for i, f in enumerate(features)], dim=1)
            return bbox, F.softmax(cls, dim=-1), ldm
    m = Neck()
    name_seeded_weights(m)
    m.eval()
    feats = retinaface_feats()
    bbox, conf, ldm = m(feats)
    cfg = {"min_sizes": [[16, 32], [64, 128], [256, 512]], "steps": [8, 16, 32], "variance": [0.1, 0.2], "clip": False}
    priors = utl.PriorBox(cfg, image_size=(128, 160)).forward()
    boxes = utl.decode(bbox[0].clone(), priors, cfg["variance"])
    lms = utl.decode_landm(ldm[0].clone(), priors, cfg["variance"])
    bb = utl.batched_decode(bbox.clone(), priors.unsqueeze(0), cfg["variance"])
    sd = m.state_dict()
    save("g11_retinaface", feat0=feats[0].half(), feat1=feats[1].half(), feat2=feats[2].half(), bbox=bbox, conf=conf, ldm=ldm,
         priors=priors, boxes=boxes, landmarks=lms, batched_boxes=bb,
         param_names=np.array(list(sd.keys())), param_shapes=np.array([";".join(map(str, v.shape)) for v in sd.values()]))


EXTRA = {"g4": g4_blocks, "g8": g8_blur_forward, "g5_64": g5_unet_64, "g9": g9_codeformer, "g10": g10_parsenet,
         "g11": g11_retinaface}


def main():
    import refimport
    refimport.install_stubs()
    torch.set_grad_enabled(False)
    only = sys.argv[1:]
    for key, fn in EXTRA.items():
        if not only or key in only:
            fn()
    if only and "base" not in only:
        return
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
