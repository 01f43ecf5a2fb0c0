"""CPU restatement of the RetinaFace (ResNet-50) face / landmark detector and of the landmark alignment that produces the
sampler's affine matrices -- SURVEY.md section 8f row 4, detection half.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functions over a STATE DICT with the reference's names, following
guided_diffusion/facelib/detection/retinaface/retinaface.py:79-156 (module tree, forward), retinaface_net.py:6-97,139-178
(conv_bn*, SSH, FPN, heads), retinaface_utils.py:8-39,254-340 (PriorBox, decode*) and
facelib/utils/face_restoration_helper.py:122-224 (get_crop_face).

Pinned: FPN / SSH / heads / PriorBox / decode / decode_landm by tests/golden/g11_retinaface.npz, generated from the reference's
own retinaface_net.py / retinaface_utils.py (tests/golden/make_golden.py g11).  PARITY UNPINNED (packages absent, no
reference-held fixture): the ResNet-50 body (torchvision.models.resnet50, restated here: v1.5 Bottleneck), torchvision.ops.nms
(greedy IoU suppression) and cv2.estimateAffinePartial2D(method=LMEDS).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _bn(sd, name, x):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        False, 0.0, 1e-5)


def conv_bn(sd, name, x, stride=1, leaky=None):
    """retinaface_net.py:6-24: Sequential(Conv2d(k, stride, k // 2, bias=False), BatchNorm2d[, LeakyReLU(leaky)])."""
    w = sd[name + ".0.weight"]
    x = _bn(sd, name + ".1", F.conv2d(x, w, None, stride, w.shape[-1] // 2))
    return F.leaky_relu(x, leaky) if leaky is not None else x


def bottleneck(sd, name, x, stride):
    """torchvision.models.resnet.Bottleneck.forward (v1.5: stride on conv2)."""
    ident = x
    out = F.relu(_bn(sd, name + ".bn1", F.conv2d(x, sd[name + ".conv1.weight"])))
    out = F.relu(_bn(sd, name + ".bn2", F.conv2d(out, sd[name + ".conv2.weight"], None, stride, 1)))
    out = _bn(sd, name + ".bn3", F.conv2d(out, sd[name + ".conv3.weight"]))
    if name + ".downsample.0.weight" in sd:
        ident = _bn(sd, name + ".downsample.1", F.conv2d(x, sd[name + ".downsample.0.weight"], None, stride))
    return F.relu(out + ident)


def resnet50_body(sd, x, prefix="body"):
    """IntermediateLayerGetter(resnet50, {'layer2': 1, 'layer3': 2, 'layer4': 3}) -- retinaface.py:99-102."""
    x = F.relu(_bn(sd, prefix + ".bn1", F.conv2d(x, sd[prefix + ".conv1.weight"], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for i, (blocks, stride) in enumerate([(3, 1), (4, 2), (6, 2), (3, 2)]):
        for b in range(blocks):
            x = bottleneck(sd, f"{prefix}.layer{i + 1}.{b}", x, stride if b == 0 else 1)
        if i >= 1:
            outs.append(x)
    return outs


def fpn(sd, feats, leaky=0):
    """retinaface_net.py:80-97."""
    o1 = conv_bn(sd, "fpn.output1", feats[0], leaky=leaky)
    o2 = conv_bn(sd, "fpn.output2", feats[1], leaky=leaky)
    o3 = conv_bn(sd, "fpn.output3", feats[2], leaky=leaky)
    o2 = conv_bn(sd, "fpn.merge2", o2 + F.interpolate(o3, size=o2.shape[2:], mode="nearest"), leaky=leaky)
    o1 = conv_bn(sd, "fpn.merge1", o1 + F.interpolate(o2, size=o1.shape[2:], mode="nearest"), leaky=leaky)
    return [o1, o2, o3]


def ssh(sd, name, x, leaky=0):
    """retinaface_net.py:52-63."""
    c3 = conv_bn(sd, name + ".conv3X3", x)
    c51 = conv_bn(sd, name + ".conv5X5_1", x, leaky=leaky)
    c5 = conv_bn(sd, name + ".conv5X5_2", c51)
    c7 = conv_bn(sd, name + ".conv7x7_3", conv_bn(sd, name + ".conv7X7_2", c51, leaky=leaky))
    return F.relu(torch.cat([c3, c5, c7], dim=1))


def head(sd, name, x, k):
    """ClassHead / BboxHead / LandmarkHead.forward -- retinaface_net.py:146-178."""
    out = F.conv2d(x, sd[name + ".conv1x1.weight"], sd[name + ".conv1x1.bias"])
    return out.permute(0, 2, 3, 1).contiguous().view(out.shape[0], -1, k)


@torch.no_grad()
def neck_and_heads(sd, feats):
    """RetinaFace.forward after the body (retinaface.py:133-156, phase 'test'): (bbox, softmax(cls), landmarks)."""
    f = fpn(sd, feats)
    feats = [ssh(sd, "ssh1", f[0]), ssh(sd, "ssh2", f[1]), ssh(sd, "ssh3", f[2])]
    bbox = torch.cat([head(sd, f"BboxHead.{i}", v, 4) for i, v in enumerate(feats)], dim=1)
    cls = torch.cat([head(sd, f"ClassHead.{i}", v, 2) for i, v in enumerate(feats)], dim=1)
    ldm = torch.cat([head(sd, f"LandmarkHead.{i}", v, 10) for i, v in enumerate(feats)], dim=1)
    return bbox, F.softmax(cls, dim=-1), ldm


@torch.no_grad()
def retinaface_forward(sd, x):
    """RetinaFace.forward (retinaface.py:129-156): x (B, 3, H, W), mean already subtracted."""
    return neck_and_heads(sd, resnet50_body(sd, x))


# ------------------------------------------------------------------------------------------ host side
CFG = {"min_sizes": [[16, 32], [64, 128], [256, 512]], "steps": [8, 16, 32], "variance": [0.1, 0.2], "clip": False}


def prior_box(image_size, cfg=CFG):
    """PriorBox.forward -- retinaface_utils.py:19-39, loop for loop."""
    from itertools import product
    from math import ceil
    anchors = []
    fmaps = [[ceil(image_size[0] / s), ceil(image_size[1] / s)] for s in cfg["steps"]]
    for k, f in enumerate(fmaps):
        for i, j in product(range(f[0]), range(f[1])):
            for min_size in cfg["min_sizes"][k]:
                s_kx = min_size / image_size[1]
                s_ky = min_size / image_size[0]
                cx = (j + 0.5) * cfg["steps"][k] / image_size[1]
                cy = (i + 0.5) * cfg["steps"][k] / image_size[0]
                anchors += [cx, cy, s_kx, s_ky]
    out = torch.Tensor(anchors).view(-1, 4)
    if cfg["clip"]:
        out.clamp_(max=1, min=0)
    return out


def decode(loc, priors, variances):
    """retinaface_utils.py:254-271."""
    boxes = torch.cat((priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:],
                       priors[:, 2:] * torch.exp(loc[:, 2:] * variances[1])), 1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def decode_landm(pre, priors, variances):
    """retinaface_utils.py:274-294."""
    return torch.cat([priors[:, :2] + pre[:, 2 * j:2 * j + 2] * variances[0] * priors[:, 2:] for j in range(5)], dim=1)


def nms(dets, thresh):
    """torchvision.ops.nms as called by py_cpu_nms (retinaface_utils.py:42-50): plain O(n^2) greedy suppression."""
    dets = np.asarray(dets, dtype=np.float32)
    order = list(np.argsort(-dets[:, 4], kind="stable"))
    keep = []
    while order:
        i = order.pop(0)
        keep.append(int(i))
        rest = []
        for j in order:
            xx1, yy1 = max(dets[i, 0], dets[j, 0]), max(dets[i, 1], dets[j, 1])
            xx2, yy2 = min(dets[i, 2], dets[j, 2]), min(dets[i, 3], dets[j, 3])
            inter = max(xx2 - xx1, 0.0) * max(yy2 - yy1, 0.0)
            a_i = (dets[i, 2] - dets[i, 0]) * (dets[i, 3] - dets[i, 1])
            a_j = (dets[j, 2] - dets[j, 0]) * (dets[j, 3] - dets[j, 1])
            if inter / (a_i + a_j - inter) <= thresh:
                rest.append(j)
        order = rest
    return keep


def batched_detect_faces(sd, frames, conf_threshold=0.8, nms_threshold=0.4):
    """RetinaFace.batched_detect_faces (retinaface.py:345-418) with use_origin_size=True: frames (B, 3, H, W) in [0, 255]."""
    mean = torch.tensor([104.0, 117.0, 123.0]).view(1, 3, 1, 1)
    loc, conf, ldm = retinaface_forward(sd, frames.float() - mean)
    H, W = frames.shape[2:]
    priors = prior_box((H, W))
    scale = torch.tensor([W, H, W, H], dtype=torch.float32)
    scale1 = torch.tensor([W, H] * 5, dtype=torch.float32)
    out = []
    for b in range(frames.shape[0]):
        boxes = decode(loc[b], priors, CFG["variance"]) * scale
        lms = decode_landm(ldm[b], priors, CFG["variance"]) * scale1
        sc = conf[b, :, 1]
        inds = sc > conf_threshold
        if not bool(inds.any()):
            continue
        dets = torch.cat((boxes[inds], sc[inds, None]), dim=1).float().numpy()
        lm = lms[inds].numpy()
        keep = nms(dets, nms_threshold)
        out.append(np.concatenate((dets[keep], lm[keep]), axis=1))
    return out


FACE_TEMPLATE_512 = np.array([[192.98138, 239.94708], [318.90277, 240.1936], [256.63416, 314.01935], [201.26117, 371.41043],
                              [313.08905, 371.15118]])          # face_restoration_helper.py:90-98


def estimate_affine_partial(src, dst):
    """cv2.estimateAffinePartial2D(src, dst, method=cv2.LMEDS)[0] restated independently of the product code: every minimal
    pair, least median of squared residuals, OpenCV's inlier threshold, closed-form similarity fit (Umeyama without
    reflection handling is not needed: the 4-parameter model is linear)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    n = len(src)

    def fit(idx):
        s, d = src[idx], dst[idx]
        ms, md = s.mean(0), d.mean(0)
        sc, dc = s - ms, d - md
        den = (sc ** 2).sum()
        a = (sc * dc).sum() / den
        b = (sc[:, 0] * dc[:, 1] - sc[:, 1] * dc[:, 0]).sum() / den
        R = np.array([[a, -b], [b, a]])
        t = md - R @ ms
        return np.concatenate([R, t[:, None]], axis=1)

    def resid(M):
        return (((src @ M[:, :2].T + M[:, 2]) - dst) ** 2).sum(1)
    best, bm = None, np.inf
    for i in range(n):
        for j in range(i + 1, n):
            if np.allclose(src[i], src[j]):
                continue
            M = fit([i, j])
            m = np.median(resid(M))
            if m < bm:
                best, bm = M, m
    if best is None:
        return None
    if n > 2 and bm > 0:
        sigma = 2.5 * 1.4826 * (1 + 5.0 / (n - 2)) * np.sqrt(bm)
        inl = np.where(resid(best) <= sigma * sigma)[0]
        if len(inl) < 2:
            inl = np.arange(n)
    else:
        inl = np.arange(n) if n == 2 or bm > 0 else np.where(resid(best) <= 1e-12)[0]
    return fit(list(inl))
