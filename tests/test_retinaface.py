"""RetinaFace detection + landmark alignment (detection half).

Pinned by tests/golden/g11_retinaface.npz (the reference's own FPN / SSH / heads / PriorBox / decode, make_golden.py g11): the
oracle and the host utilities on the CPU, the HIP module on the GPU.  PARITY UNPINNED, tested for self-consistency only: the
ResNet-50 body (torchvision), NMS (torchvision.ops.nms) and the LMedS partial-affine estimate (cv2)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.weights import name_seeded_weights

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g11_retinaface.npz")
VAR = [0.1, 0.2]


def _model(device="cpu", tame=False):
    """tame: box / landmark regressions scaled down so that decoded boxes of the random-weight network stay inside a few image
    sizes (exp(0.2 * loc) of unit-variance regressions on 50-layer random features overflows) -- for the detection pipeline tests."""
    from flair_amd.guided_diffusion.retinaface import RetinaFace
    m = RetinaFace(network_name="resnet50", device="cpu")
    name_seeded_weights(m)
    if tame:
        with torch.no_grad():
            for n, p_ in m.named_parameters():
                if n.startswith(("BboxHead", "LandmarkHead")):
                    p_.mul_(0.01)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    if device != "cpu":
        m = m.to(device)
        m.device = torch.device(device)
    return m.eval(), sd


def _feats(g):
    return [torch.from_numpy(g[f"feat{i}"]).float() for i in range(3)]


# ------------------------------------------------------------------------------------------------------ CPU
def test_neck_and_head_names_match_reference():
    g = np.load(GOLD)
    _, sd = _model()
    names = [str(n) for n in g["param_names"]]
    shapes = {str(n): str(s) for n, s in zip(g["param_names"], g["param_shapes"])}
    assert all(n in sd for n in names), [n for n in names if n not in sd][:5]
    assert all(";".join(map(str, sd[n].shape)) == shapes[n] for n in names)
    # everything else is the ResNet-50 body behind IntermediateLayerGetter: conv1 / bn1 / layer1..4, no fc
    rest = [k for k in sd if k not in names]
    assert all(k.startswith("body.") for k in rest) and not any("fc" in k for k in rest)
    # torchvision resnet50 without its fc: 23 508 032 parameters + 53 120 BatchNorm running statistics in 318 entries
    assert len(rest) == 318 and sum(sd[k].numel() for k in rest if "num_batches" not in k) == 23_508_032 + 53_120


def test_oracle_neck_and_heads_match_reference_fixture():
    from oracle import retinaface as orf
    g = np.load(GOLD)
    _, sd = _model()
    bbox, conf, ldm = orf.neck_and_heads(sd, _feats(g))
    for got, key in ((bbox, "bbox"), (conf, "conf"), (ldm, "ldm")):
        ref = torch.from_numpy(g[key])
        assert torch.allclose(got, ref, atol=2e-5 * max(1.0, ref.abs().max().item()), rtol=0), key


def test_priors_and_decoding_match_reference_fixture():
    from flair_amd.guided_diffusion import retinaface_utils as ru
    from oracle import retinaface as orf
    g = np.load(GOLD)
    cfg = {"min_sizes": [[16, 32], [64, 128], [256, 512]], "steps": [8, 16, 32], "clip": False}
    pri = ru.PriorBox(cfg, image_size=(128, 160)).forward()
    assert np.array_equal(pri, g["priors"])                                   # bit-exact: double arithmetic, one rounding
    assert np.array_equal(orf.prior_box((128, 160)).numpy(), g["priors"])
    boxes = ru.decode(g["bbox"][0], pri, VAR)
    lms = ru.decode_landm(g["ldm"][0], pri, VAR)
    assert np.abs(boxes - g["boxes"]).max() <= 2e-6 * np.abs(g["boxes"]).max()
    assert np.abs(lms - g["landmarks"]).max() <= 2e-6 * np.abs(g["landmarks"]).max()
    assert np.abs(ru.batched_decode(g["bbox"], pri[None], VAR) - g["batched_boxes"]).max() <= 2e-6 * np.abs(g["boxes"]).max()
    assert np.abs(ru.batched_decode_landm(g["ldm"], pri[None], VAR)[0] - g["landmarks"]).max() <= 2e-6 * np.abs(g["landmarks"]).max()
    ob = orf.decode(torch.from_numpy(g["bbox"][0]), torch.from_numpy(g["priors"]), VAR).numpy()
    assert np.abs(ob - g["boxes"]).max() <= 1e-6 * np.abs(g["boxes"]).max()


def test_nms_against_the_plain_algorithm_and_known_cases():
    from flair_amd.guided_diffusion.retinaface_utils import py_cpu_nms
    from oracle.retinaface import nms
    rng = np.random.default_rng(3)
    for n in (0, 1, 7, 300):
        xy = rng.uniform(0, 100, (n, 2))
        dets = np.concatenate([xy, xy + rng.uniform(4, 50, (n, 2)), rng.uniform(size=(n, 1))], axis=1).astype(np.float32)
        assert py_cpu_nms(dets, 0.4) == (nms(dets, 0.4) if n else [])
    # two heavily overlapping boxes and a separate one: the better of the pair and the separate one survive, best first
    dets = np.array([[0, 0, 10, 10, 0.6], [1, 1, 11, 11, 0.9], [50, 50, 60, 60, 0.7]], dtype=np.float32)
    assert py_cpu_nms(dets, 0.4) == [1, 2]
    assert py_cpu_nms(dets, 0.95) == [1, 2, 0]


def test_partial_affine_estimate():
    from flair_amd.guided_diffusion.retinaface_utils import estimate_affine_partial
    from oracle.retinaface import FACE_TEMPLATE_512, estimate_affine_partial as oracle_est
    rng = np.random.default_rng(5)
    tpl = FACE_TEMPLATE_512
    th, sc, t = -0.21, 0.37, np.array([91.0, 40.5])
    R = sc * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    lm = (tpl - np.array([256.0, 300.0])) @ R.T + t                           # landmarks = a similarity of the template
    Rinv = np.linalg.inv(R)
    M = estimate_affine_partial(lm, tpl)                                      # maps landmarks back onto the template
    assert np.abs(M[:, :2] - Rinv).max() < 1e-9
    assert np.abs(lm @ M[:, :2].T + M[:, 2] - tpl).max() < 1e-8
    # 4-DOF model: equal diagonal, opposite off-diagonal
    assert abs(M[0, 0] - M[1, 1]) < 1e-12 and abs(M[0, 1] + M[1, 0]) < 1e-12
    # one gross outlier (a landmark on the wrong feature) does not move the estimate; noise on all five is averaged
    bad = lm.copy()
    bad[4] += [35.0, -20.0]
    assert np.abs(estimate_affine_partial(bad, tpl)[:, :2] - Rinv).max() < 1e-9
    noisy = lm + rng.normal(scale=0.3, size=lm.shape)
    Mn = estimate_affine_partial(noisy, tpl)
    assert np.abs(noisy @ Mn[:, :2].T + Mn[:, 2] - tpl).max() < 4.0
    for pts in (lm, bad, noisy, lm[:3]):
        a, b = estimate_affine_partial(pts, tpl[:len(pts)]), oracle_est(pts, tpl[:len(pts)])
        assert np.abs(a - b).max() < 1e-8 * max(1.0, np.abs(b).max())
    assert estimate_affine_partial(lm[:1], tpl[:1]) is None


# ------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_add_act_and_maxpool_kernels(dev, dtype):
    import torch.nn.functional as F
    from flair_amd import ops
    from tests.util import from_clip, rb, to_clip
    g = torch.Generator().manual_seed(2)
    x = rb(torch.randn(2, 64, 13, 18, generator=g), dtype)
    y = rb(torch.randn(2, 64, 13, 18, generator=g), dtype)
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    for act, fn in ((ops.ACT_NONE, lambda v: v), (ops.ACT_RELU, torch.relu), (ops.ACT_LRELU01, lambda v: F.leaky_relu(v, 0.1))):
        got = from_clip(ops.add_act(to_clip(x, dtype, dev), to_clip(y, dtype, dev), act))
        assert (got - fn(x + y)).abs().max().item() <= tol * 4
    got = from_clip(ops.add_act(to_clip(x, dtype, dev), None, ops.ACT_RELU))
    assert (got - torch.relu(x)).abs().max().item() == 0
    for hw in ((13, 18), (16, 16), (7, 5)):
        xs = x[..., :hw[0], :hw[1]].contiguous()
        got = from_clip(ops.maxpool3x3s2(to_clip(xs, dtype, dev)))
        assert torch.equal(got, F.max_pool2d(xs, 3, 2, 1))


@pytest.mark.gpu
def test_hip_neck_and_heads_match_reference_fixture(dev):
    from tests.util import to_clip
    g = np.load(GOLD)
    m, _ = _model(dev)
    m._ensure_packed(dev)
    bbox, cls, ldm = m._neck_heads([to_clip(f, torch.float32, dev) for f in _feats(g)])
    conf = torch.softmax(cls, dim=-1)
    for got, key in ((bbox, "bbox"), (conf, "conf"), (ldm, "ldm")):
        ref = torch.from_numpy(g[key])
        err = (got.cpu() - ref).abs().max().item()
        assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (key, err)


@pytest.mark.gpu
def test_hip_retinaface_matches_oracle_end_to_end(dev):
    """The whole detector (ResNet-50 body + FPN + SSH + heads) on two 128 x 160 frames against the CPU oracle, then the
    detections of the full host pipeline (priors, decoding, threshold, NMS)."""
    from oracle import retinaface as orf
    m, sd = _model(dev, tame=True)
    g = torch.Generator().manual_seed(8)
    frames = torch.rand(2, 3, 128, 160, generator=g) * 255.0
    mean = torch.tensor([104.0, 117.0, 123.0]).view(1, 3, 1, 1)
    ref = orf.retinaface_forward(sd, frames - mean)
    got = m((frames - mean).to(dev))
    for a, b, key in zip(got, ref, ("bbox", "conf", "ldm")):
        err = (a.cpu() - b).abs().max().item()
        assert err <= 1e-3 * max(1.0, b.abs().max().item()), (key, err)
    # detections: random weights score around 0.5, so a threshold of 0.5 keeps hundreds of boxes
    thr = 0.5
    dets = m.batched_detect_faces(frames.to(dev), thr)
    odets = orf.batched_detect_faces(sd, frames, thr)
    assert len(dets) == len(odets) == 2
    for d, o in zip(dets, odets):
        # boxes whose score sits within 1e-3 of the threshold may be kept by one side only; everything else must agree
        assert abs(len(d) - len(o)) <= max(3, len(o) // 50), (len(d), len(o))
        match = 0
        for row in o:
            j = np.argmin(np.abs(d[:, :5] - row[:5]).sum(axis=1))
            match += np.abs(d[j] - row).max() <= 1e-2 * max(1.0, np.abs(row).max())
        assert match >= 0.95 * len(o), (match, len(o))


@pytest.mark.gpu
def test_helper_get_crop_face_runs_detector_alignment_and_crop(dev):
    """FaceRestoreHelper.get_crop_face without an external detector: RetinaFace on the GPU, landmark alignment on the host, crop
    by flair_warp_affine_cubic -- against the same pipeline assembled from the CPU oracles."""
    from flair_amd.guided_diffusion.face_restoration_helper import FaceRestoreHelper
    from oracle import facewarp as ofw
    from oracle import retinaface as orf
    m, sd = _model(dev, tame=True)
    helper = FaceRestoreHelper(face_size=128, device=dev, face_det=m)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
    faces, mats, idx = helper.get_crop_face(x.to(dev), only_center_face=True)
    assert faces is not None and tuple(faces.shape) == (len(idx), 3, 128, 128) and idx == [0, 1]
    frames255 = ((x + 1) / 2).clamp(0, 1) * 255
    odets = orf.batched_detect_faces(sd, frames255, 0.5)
    tpl = orf.FACE_TEMPLATE_512 * (128 / 512.0)
    for k, (bboxes, M) in enumerate(zip(odets, mats)):
        cx = [np.linalg.norm([(b[0] + b[2]) / 2 - 64, (b[1] + b[3]) / 2 - 64]) for b in bboxes]
        b = bboxes[int(np.argmin(cx))]
        Mo = orf.estimate_affine_partial(np.array([[b[i], b[i + 1]] for i in range(5, 15, 2)]), tpl)
        assert np.abs(M - Mo).max() <= 2e-2 * max(1.0, np.abs(Mo).max()), (k, M, Mo)
        img = frames255[k].permute(1, 2, 0).numpy().astype(np.float32)
        crop = ofw.warp_affine_cubic(img, M, (128, 128), border=(135.0, 133.0, 132.0))
        ref = torch.from_numpy(((crop / 255.0 - 0.5) / 0.5).clip(-1, 1)).permute(2, 0, 1)
        assert (faces[k].cpu() - ref).abs().max().item() <= 2e-3


@pytest.mark.gpu
def test_helper_get_crop_face_selection_rules_with_fixed_detections(dev):
    """The host logic of get_crop_face (face_restoration_helper.py:122-224) on detections handed in by a stub detector: the first /
    largest / centre face, the eye-distance filter, frames without a detection dropped from the result, the template offsets and
    scale, and the crop = get_crop_face_from_affine_matrices of the chosen matrices."""
    from flair_amd.guided_diffusion.face_restoration_helper import FaceRestoreHelper
    from flair_amd.guided_diffusion.retinaface_utils import estimate_affine_partial

    def face(cx, cy, size, score):
        # box + five landmarks of a face of the given size centred at (cx, cy): the 512-template scaled and shifted
        tpl = np.array([[192.98138, 239.94708], [318.90277, 240.1936], [256.63416, 314.01935], [201.26117, 371.41043],
                        [313.08905, 371.15118]]) / 512.0 - 0.5
        lm = tpl * size + np.array([cx, cy])
        return np.concatenate([[cx - size / 2, cy - size / 2, cx + size / 2, cy + size / 2, score], lm.reshape(-1)]).astype(np.float32)

    small_first = face(30.0, 34.0, 24.0, 0.99)          # first in the list, small, far from the centre
    large = face(84.0, 40.0, 56.0, 0.9)                 # the largest box
    central = face(66.0, 62.0, 30.0, 0.8)               # nearest the centre of a 128 x 128 frame
    per_frame = [np.stack([small_first, large, central]), np.zeros((0, 15), np.float32), np.stack([central])]

    class StubDetector:
        def batched_detect_faces(self, frames, conf_threshold=0.8, nms_threshold=0.4, use_origin_size=True, pre=None):
            assert tuple(frames.shape) == (3, 3, 128, 128) and conf_threshold == 0.5 and pre == (127.5, 127.5, 0.0, 255.0)
            return [d for d in per_frame if len(d)]      # the reference's detector skips frames without a face (retinaface.py:393-395)

    helper = FaceRestoreHelper(face_size=128, device=dev, face_det=StubDetector())
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(3, 3, 128, 128, generator=g) * 2 - 1).to(dev)
    tpl = helper.face_template

    def lms(row):
        return row[5:15].reshape(5, 2)
    for kw, want in ((dict(), small_first), (dict(only_keep_largest=True), large), (dict(only_center_face=True), central)):
        faces, mats, idx = helper.get_crop_face(x, **kw)
        assert idx == [0, 1] and len(mats) == 2            # two results: the detector's list pairs with the frames in order
        assert np.allclose(mats[0], estimate_affine_partial(lms(want), tpl))
        assert np.allclose(mats[1], estimate_affine_partial(lms(central), tpl))
        ref = helper.get_crop_face_from_affine_matrices(x[idx].contiguous(), mats)
        assert torch.equal(faces, ref) and tuple(faces.shape) == (2, 3, 128, 128)
    # eye distance: small_first's eyes are 24 * 0.246 = 5.9 px apart, large's 13.8, central's 7.4
    faces, mats, idx = helper.get_crop_face(x, eye_dist_threshold=7.0)
    assert np.allclose(mats[0], estimate_affine_partial(lms(large), tpl))
    faces, mats, idx = helper.get_crop_face(x, eye_dist_threshold=20.0)
    assert faces is None and mats is None and idx is None
    # template scale and offsets
    faces, mats, idx = helper.get_crop_face(x, only_center_face=True, face_template_resize=0.5, face_template_x_offset=4.0, face_template_y_offset=-2.0)
    assert np.allclose(mats[0], estimate_affine_partial(lms(central), (tpl + np.array([4.0, -2.0])) * 0.5))
    with pytest.raises(NotImplementedError):
        helper.get_crop_face(x, resize=256)
