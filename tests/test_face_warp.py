"""Un-aligned prior branch: device-side face crop / inverse paste (second half;
gaussian_diffusion.py:476-493 -> facelib/utils/face_restoration_helper.py:225-254, 264-335).

**Parity unpinned**: the reference calls cv2.warpAffine / cv2.GaussianBlur, cv2 is not installable here and the
reference holds no fixture of these calls.  oracle/facewarp.py restates OpenCV 4.4's published algorithms in numpy
(checked below against closed forms and scipy where those exist); the HIP kernels are compared with that oracle."""
import numpy as np
import pytest
import torch


def _similarity(scale, theta, tx, ty):
    c, s = scale * np.cos(theta), scale * np.sin(theta)
    return np.array([[c, -s, tx], [s, c, ty]], dtype=np.float64)


# ------------------------------------------------------------------------------------- oracle sanity (CPU)
def test_oracle_warp_closed_forms():
    from oracle import facewarp as fw
    rng = np.random.default_rng(0)
    img = (rng.random((48, 64, 3)) * 255).astype(np.float32)
    eye = np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float64)
    assert np.array_equal(fw.warp_affine_cubic(img, eye, (64, 48)), img)          # zero fraction: weights (0, 1, 0, 0)
    sh = fw.warp_affine_cubic(img, np.array([[1, 0, 5], [0, 1, -3]], dtype=np.float64), (64, 48), border=(1, 2, 3))
    assert np.array_equal(sh[10:40, 10:60], img[13:43, 5:55])                     # integer shift is a copy
    assert np.array_equal(sh[0, 0], np.array([1, 2, 3], dtype=np.float32))        # fully outside -> border value
    # a smooth image under a similarity transform follows the analytic mapping (cubic + 1/32-pixel quantisation error)
    yy, xx = np.mgrid[0:48, 0:64]
    sm = (np.sin(xx / 9.0) + np.cos(yy / 7.0)).astype(np.float64)
    M = _similarity(1.2, 0.3, 4.0, -6.0)
    out = fw.warp_affine_cubic(sm, M, (64, 48))
    Mi = fw.invert_affine(M)
    X = Mi[0, 0] * xx + Mi[0, 1] * yy + Mi[0, 2]
    Y = Mi[1, 0] * xx + Mi[1, 1] * yy + Mi[1, 2]
    inside = (X > 2) & (X < 60) & (Y > 2) & (Y < 44)
    assert np.abs(out - (np.sin(X / 9.0) + np.cos(Y / 7.0)))[inside].max() < 3e-2
    assert np.allclose(fw.invert_affine(fw.invert_affine(M)), M, atol=1e-12)
    # the 32-entry weight table: rows sum to 1, fraction 0 is the identity tap
    assert np.allclose(fw._TAB.sum(1), 1.0, atol=1e-6) and np.array_equal(fw._TAB[0], np.array([0, 1, 0, 0], dtype=np.float32))


def test_oracle_blur_matches_scipy():
    from scipy.ndimage import gaussian_filter1d
    from oracle import facewarp as fw
    rng = np.random.default_rng(1)
    m = rng.random((96, 130)) * 255
    got = fw.gaussian_blur(m, 101, 26)
    tr = 50 / 26 + 1e-9                                       # radius 50 = the 101-tap kernel
    ref = gaussian_filter1d(gaussian_filter1d(m, 26, axis=1, mode="mirror", truncate=tr), 26, axis=0, mode="mirror", truncate=tr)
    assert np.abs(got - ref).max() < 1e-10
    k = fw.gaussian_kernel(101, 26.0)
    assert abs(k.sum() - 1) < 1e-15 and np.array_equal(k, k[::-1])


def test_helper_refuses_unbuilt_detectors_and_cpu_tensors():
    from flair_amd._lib import FlairHipError
    from flair_amd.guided_diffusion.face_restoration_helper import FaceRestoreHelper, invert_affine
    from oracle import facewarp as fw
    h = FaceRestoreHelper(device="cpu", det_model="YOLOv5l")
    with pytest.raises(NotImplementedError):              # only retinaface_resnet50 is built
        h.get_crop_face(torch.zeros(1, 3, 64, 64))
    h = FaceRestoreHelper(device="cpu")
    with pytest.raises(NotImplementedError):              # resize=...: the reference's own expression raises there
        h.get_crop_face(torch.zeros(1, 3, 64, 64), resize=32)
    with pytest.raises(FlairHipError):                    # the detector has no CPU path
        h.get_crop_face(torch.zeros(1, 3, 64, 64))
    assert h.get_crop_face_from_affine_matrices(torch.zeros(0, 3, 8, 8), []) is None
    M = _similarity(1.7, -0.2, 3.0, 9.0)
    assert np.array_equal(invert_affine(M), fw.invert_affine(M))     # host glue == oracle restatement


# --------------------------------------------------------------------------------------------- GPU vs oracle
def _frames(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.interpolate(torch.randn(B, 3, S // 8, S // 8, generator=g), size=(S, S), mode="bicubic",
                                           align_corners=False)
    return (base * 0.6 + 0.15 * torch.randn(B, 3, S, S, generator=g)).clamp(-1.3, 1.3)       # some values beyond [-1, 1]


def _matrices(B, S):
    # similarity transforms as estimateAffinePartial2D returns them (frame -> 512x512 template): faces of different size /
    # tilt / position, one partly outside the frame (border fill in the crop, zero fill in the paste)
    ms = [_similarity(1.45, 0.08, -96.3, -71.8), _similarity(2.1, -0.21, -330.2, -260.4), _similarity(0.93, 0.3, 61.0, -20.5),
          _similarity(1.0, 0.0, 0.0, 0.0)]
    return [ms[i % len(ms)] for i in range(B)]


@pytest.mark.gpu
def test_crop_faces_vs_oracle(dev):
    from flair_amd.guided_diffusion.face_restoration_helper import FaceRestoreHelper
    from oracle import facewarp as fw
    B, S = 4, 512
    x = _frames(B, S, 3)
    mats = _matrices(B, S)
    ref = fw.get_crop_face_from_affine_matrices(x, mats)
    got = FaceRestoreHelper(device=dev).get_crop_face_from_affine_matrices(x.to(dev), mats)
    torch.cuda.synchronize()
    assert got.shape == ref.shape == (B, 3, 512, 512)
    err = (got.cpu() - ref).abs().max().item()
    assert err <= 2e-6, err                                  # same f32 arithmetic in the same order (last-bit differences of the division)
    # the identity matrix crops the frame itself (up to the 8-bit-style scaling round trip in f32)
    assert (got[3].cpu() - x[3].clamp(-1, 1)).abs().max().item() <= 1e-6


@pytest.mark.gpu
def test_mask_blur_and_inverse_warp_vs_oracle(dev):
    """inverse_faces from a GIVEN parsing map: colour map, two float64 blurs, border, /255, inverse cubic warps of face and mask."""
    from flair_amd import ops
    from flair_amd.guided_diffusion import face_restoration_helper as frh
    from oracle import facewarp as fw
    B, S = 3, 512
    faces = _frames(B, S, 5)
    mats = _matrices(B, S)
    g = torch.Generator().manual_seed(6)
    blob = torch.nn.functional.interpolate(torch.rand(B, 1, 16, 16, generator=g), size=(S, S), mode="bilinear")
    parse = (blob[:, 0] * 19).long().clamp(0, 18)             # blobby label map over all 19 classes
    ref_faces, ref_masks = fw.inverse_faces(faces, mats, parse.numpy())
    helper = frh.FaceRestoreHelper(device=dev)
    lut, kern = helper._consts(dev)
    mask = ops.face_mask_blur(parse.to(torch.int32).reshape(-1).contiguous().to(dev), B, S, S, lut, kern)
    # the blurred float64 mask itself (before the warp)
    cm = np.asarray(fw.MASK_COLORMAP, dtype=np.float64)
    m0 = fw.gaussian_blur(fw.gaussian_blur(cm[parse[0].numpy()], 101, 26), 101, 26)
    m0[:10] = 0; m0[-10:] = 0; m0[:, :10] = 0; m0[:, -10:] = 0
    assert np.abs(mask[0, 0].cpu().numpy() - m0 / 255.0).max() <= 1e-12
    minv = helper._minv(helper.get_inverse_affine(mats), dev)
    inv_masks = ops.warp_affine_cubic(mask, minv, (S, S))
    inv_faces = ops.warp_affine_cubic(faces.to(dev).contiguous(), minv, (S, S), pre=True, post=True)
    torch.cuda.synchronize()
    assert (inv_masks.cpu() - ref_masks).abs().max().item() <= 1e-6
    assert (inv_faces.cpu() - ref_faces).abs().max().item() <= 2e-6


@pytest.mark.gpu
def test_unaligned_sampler_steps_vs_oracle(dev):
    """Two sampler steps with aligned=False at the reference's 512x512 (gaussian_diffusion.py:465-515): toy network, HIP
    ParseNet + HIP crop / paste against the oracle loop driven by oracle/facewarp.py + oracle/parsenet.py.  The parsing
    arg-max of two f32 implementations may differ on near-ties; two 101-tap sigma-26 blurs spread a flipped pixel to
    < 1e-4 of mask, hence the 2e-3 bound on x0."""
    from flair_amd import workload as wl
    from flair_amd.guided_diffusion.face_restoration_helper import FaceRestoreHelper
    from flair_amd.guided_diffusion.parsenet import ParseNet
    from oracle import diffusion as odiff
    from oracle import facewarp as fw
    from oracle import parsenet as opn
    from tests.golden.weights import name_seeded_weights
    from tests.test_gpu_sampler import toy_model
    T, S, STEPS = 2, 512, 10
    net = name_seeded_weights(ParseNet(in_size=512, out_size=512, parsing_ch=19)).eval()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    mats = _matrices(T, S)[:T]
    g = torch.Generator().manual_seed(11)
    x_T = torch.randn(T, 3, S, S, generator=g)
    tape = [torch.randn(T, 3, S, S, generator=g) for _ in range(2)]
    aux = lambda face, t, xt: 0.85 * face + 0.05 * xt         # noqa: E731  (stand-in prior on the CROPS)

    class OracleHelper:
        def get_crop_face_from_affine_matrices(self, imgs, ms):
            return fw.get_crop_face_from_affine_matrices(imgs, ms)

        def inverse_faces(self, restored, ms):
            with torch.no_grad():
                parse = opn.parsenet_forward(sd, restored)[0].argmax(1)
            return fw.inverse_faces(restored, ms, parse.numpy())
    tab = odiff.Spaced(odiff.spaced_steps(1000, str(STEPS)), odiff.named_betas("face_blur", 1000))
    ref_trace, calls = [], []

    class Stop(Exception):
        pass

    def omodel(x, t, **kw):
        if len(calls) == 2:
            raise Stop()
        calls.append(1)
        return toy_model(x, t, **kw)
    try:
        odiff.sample_loop(tab, omodel, x_T, model_kwargs=dict(num_frames=T), aux_model=aux, w=0.5, tau=2, rho=0.25,
                          step_noise=tape + tape, trace=ref_trace, aligned=False, face_restore_helper=OracleHelper(),
                          affine_matrices=mats)
    except Stop:
        pass
    assert len(ref_trace) == 2

    class M:
        def parameters(self):
            return iter([x_T.to(dev)])

        def __call__(self, x, t, **kw):
            return toy_model(x, t, **kw)
    diffusion = wl.diffusion_for(STEPS)
    gen = diffusion.p_sample_loop_progressive(
        M(), x_T.shape, noise=x_T.to(dev), model_kwargs=dict(num_frames=T), device=dev, aux_model=aux, w=0.5, tau=2,
        aligned=False, rho=0.25, face_restore_helper=FaceRestoreHelper(device=dev, face_parse=net), affine_matrices=mats,
        noise_fn=lambda it, like: tape[it].to(dev))
    for (ti, x0r, sr) in ref_trace:
        out = next(gen)
        assert int(out["t"][0]) == ti
        assert (out["pred_xstart"].cpu() - x0r).abs().max().item() <= 2e-3
        assert (out["sample"].cpu() - sr).abs().max().item() <= 2e-3 * max(1.0, sr.abs().max().item())
    # aligned=False without the helper / matrices is an error, not a silent aligned run
    with pytest.raises(ValueError):
        next(diffusion.p_sample_loop_progressive(M(), x_T.shape, noise=x_T.to(dev), model_kwargs=dict(num_frames=T), device=dev,
                                                 aux_model=aux, aligned=False))
