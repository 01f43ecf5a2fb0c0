"""Host-side pieces of the RetinaFace detector and of the landmark alignment: priors, box / landmark decoding, NMS,
partial-affine estimation.

Mirror of ``guided_diffusion/facelib/detection/retinaface/retinaface_utils.py`` (``PriorBox`` :8-39, ``py_cpu_nms``
:42-50, ``decode`` :254-271, ``decode_landm`` :274-294, ``batched_decode`` :297-317, ``batched_decode_landm`` :320-340) --
same names and argument meaning, numpy float32 instead of torch tensors: the reference moves every one of these results to
the host right away (``retinaface.py:239-262,383-409``), and they run once per 10-frame window.  ``estimate_affine_partial``
stands in for ``cv2.estimateAffinePartial2D(landmark, template, method=cv2.LMEDS)``
(``facelib/utils/face_restoration_helper.py:197-199``): cv2 is not a dependency here.
"""
from math import ceil

import numpy as np


class PriorBox(object):
    """Anchor centres / sizes in image-relative units, (sum_k H_k W_k len(min_sizes[k]), 4) float32 rows (cx, cy, w, h)
    in the reference's order: level, then row, then column, then size."""

    def __init__(self, cfg, image_size=None, phase="train"):
        self.min_sizes = cfg["min_sizes"]
        self.steps = cfg["steps"]
        self.clip = cfg["clip"]
        self.image_size = image_size
        self.feature_maps = [[ceil(self.image_size[0] / step), ceil(self.image_size[1] / step)] for step in self.steps]
        self.name = "s"

    def forward(self):
        out = []
        H, W = float(self.image_size[0]), float(self.image_size[1])
        for k, (fh, fw) in enumerate(self.feature_maps):
            # the reference evaluates these in Python floats (double) and rounds once, in torch.Tensor(anchors)
            cy = (np.arange(fh, dtype=np.float64) + 0.5) * self.steps[k] / H
            cx = (np.arange(fw, dtype=np.float64) + 0.5) * self.steps[k] / W
            lvl = np.empty((fh, fw, len(self.min_sizes[k]), 4), dtype=np.float64)
            lvl[..., 0] = cx[None, :, None]
            lvl[..., 1] = cy[:, None, None]
            for m, ms in enumerate(self.min_sizes[k]):
                lvl[:, :, m, 2] = ms / W
                lvl[:, :, m, 3] = ms / H
            out.append(lvl.reshape(-1, 4))
        output = np.concatenate(out, axis=0).astype(np.float32)
        if self.clip:
            np.clip(output, 0, 1, out=output)
        return output


def py_cpu_nms(dets, thresh):
    """Greedy IoU suppression, the algorithm of ``torchvision.ops.nms`` (what the reference's py_cpu_nms calls): boxes in
    descending score order (stable), a box is dropped when its IoU with an already kept box exceeds ``thresh``.
    dets: (n, 5) [x1, y1, x2, y2, score].  Returns the kept indices, best first."""
    dets = np.asarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return []
    x1, y1, x2, y2, sc = (dets[:, i] for i in range(5))
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-sc, kind="stable")
    keep = []
    suppressed = np.zeros(dets.shape[0], dtype=bool)
    for _i, i in enumerate(order):
        if suppressed[i]:
            continue
        keep.append(int(i))
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        inter = np.maximum(xx2 - xx1, 0) * np.maximum(yy2 - yy1, 0)
        iou = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[iou > thresh]] = True
    return keep


def decode(loc, priors, variances):
    loc, priors = np.asarray(loc, dtype=np.float32), np.asarray(priors, dtype=np.float32)
    boxes = np.concatenate((priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:],
                            priors[:, 2:] * np.exp(loc[:, 2:] * variances[1])), axis=1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def decode_landm(pre, priors, variances):
    pre, priors = np.asarray(pre, dtype=np.float32), np.asarray(priors, dtype=np.float32)
    return np.concatenate([priors[:, :2] + pre[:, 2 * j:2 * j + 2] * variances[0] * priors[:, 2:] for j in range(5)], axis=1)


def batched_decode(b_loc, priors, variances):
    b_loc, priors = np.asarray(b_loc, dtype=np.float32), np.asarray(priors, dtype=np.float32)
    boxes = np.concatenate((priors[:, :, :2] + b_loc[:, :, :2] * variances[0] * priors[:, :, 2:],
                            priors[:, :, 2:] * np.exp(b_loc[:, :, 2:] * variances[1])), axis=2)
    boxes[:, :, :2] -= boxes[:, :, 2:] / 2
    boxes[:, :, 2:] += boxes[:, :, :2]
    return boxes


def batched_decode_landm(pre, priors, variances):
    pre, priors = np.asarray(pre, dtype=np.float32), np.asarray(priors, dtype=np.float32)
    return np.concatenate([priors[:, :, :2] + pre[:, :, 2 * j:2 * j + 2] * variances[0] * priors[:, :, 2:] for j in range(5)],
                          axis=2)


# ----------------------------------------------------------------------------------------- landmark alignment
def _similarity_from_pairs(src, dst):
    """Least-squares 4-DOF similarity (uniform scale, rotation, translation) dst ~ [[a, -b, tx], [b, a, ty]] [src; 1]:
    the model of cv2.estimateAffinePartial2D; exact for two point pairs."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    n = src.shape[0]
    A = np.zeros((2 * n, 4))
    A[0::2, 0], A[0::2, 1], A[0::2, 2] = src[:, 0], -src[:, 1], 1.0
    A[1::2, 0], A[1::2, 1], A[1::2, 3] = src[:, 1], src[:, 0], 1.0
    sol, *_ = np.linalg.lstsq(A, dst.reshape(-1), rcond=None)
    a, b, tx, ty = sol
    return np.array([[a, -b, tx], [b, a, ty]], dtype=np.float64)


def estimate_affine_partial(src, dst, confidence=0.99):
    """``cv2.estimateAffinePartial2D(src, dst, method=cv2.LMEDS)[0]``, restated.

    OpenCV (calib3d ptsetreg.cpp, LMeDSPointSetRegistrator + AffinePartial2DEstimatorCallback) draws random minimal samples
    of 2 point pairs, keeps the model with the smallest MEDIAN squared residual, marks as inliers the pairs within
    2.5 * 1.4826 * (1 + 5 / (n - 2)) * sqrt(median) and refines the 4 parameters on the inliers by Levenberg-Marquardt.
    With the 5 (or 3) landmark pairs of this path there are only C(n, 2) <= 10 minimal samples: this function evaluates
    ALL of them (the limit of the random sampler, and deterministic), and refines by linear least squares, which is the
    minimum the LM iteration converges to for this model (linear in a, b, tx, ty).  PARITY UNPINNED: cv2 is not installable
    in the build container and the reference holds no fixture of this call; where OpenCV's random sampler misses the best
    pair, or ties in the median are broken differently, the two can differ.
    Returns a (2, 3) float64 matrix, or None for fewer than 2 pairs / a degenerate configuration."""
    src, dst = np.asarray(src, dtype=np.float64).reshape(-1, 2), np.asarray(dst, dtype=np.float64).reshape(-1, 2)
    n = src.shape[0]
    if n < 2:
        return None
    best, best_med = None, np.inf
    for i in range(n):
        for j in range(i + 1, n):
            if np.allclose(src[i], src[j]):
                continue
            M = _similarity_from_pairs(src[[i, j]], dst[[i, j]])
            err = np.sum((src @ M[:, :2].T + M[:, 2] - dst) ** 2, axis=1)
            med = np.median(err)
            if med < best_med:
                best, best_med = M, med
    if best is None:
        return None
    if n > 2:
        sigma = 2.5 * 1.4826 * (1.0 + 5.0 / (n - 2)) * np.sqrt(best_med)
        err = np.sum((src @ best[:, :2].T + best[:, 2] - dst) ** 2, axis=1)
        inl = err <= max(sigma * sigma, 1e-12) if best_med > 0 else err <= 1e-12
        if inl.sum() < 2:
            inl = np.ones(n, dtype=bool)
    else:
        inl = np.ones(n, dtype=bool)
    return _similarity_from_pairs(src[inl], dst[inl])


def get_largest_face(det_faces, h, w):
    """facelib/utils/face_restoration_helper.py:27-45: the detection with the largest box area after clamping every
    coordinate to the image."""
    def loc(v, length):
        return 0 if v < 0 else (length if v > length else v)
    areas = [(loc(d[2], w) - loc(d[0], w)) * (loc(d[3], h) - loc(d[1], h)) for d in det_faces]
    i = areas.index(max(areas))
    return det_faces[i], i


def get_center_face(det_faces, h=0, w=0, center=None):
    """facelib/utils/face_restoration_helper.py:48-62: the detection whose box centre is nearest the image centre."""
    c = np.array(center) if center is not None else np.array([w / 2, h / 2])
    dist = [np.linalg.norm(np.array([(d[0] + d[2]) / 2, (d[1] + d[3]) / 2]) - c) for d in det_faces]
    i = dist.index(min(dist))
    return det_faces[i], i
