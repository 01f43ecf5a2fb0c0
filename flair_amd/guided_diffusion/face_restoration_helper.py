"""Device-side face crop / inverse paste for the sampler's un-aligned prior branch.

Mirror of the two methods of the reference's ``FaceRestoreHelper`` that the sampling loop calls every denoising step
(``guided_diffusion/facelib/utils/face_restoration_helper.py:225-254`` ``get_crop_face_from_affine_matrices`` and
``:264-335`` ``inverse_faces``, called at ``gaussian_diffusion.py:476-493``), same names, argument meaning and return
values -- but the frames stay in HBM: the reference converts every frame to numpy, runs ``cv2.warpAffine`` /
``cv2.GaussianBlur`` on the host and uploads the result, twice per step.

The affine matrices are an INPUT: the reference estimates them once per window (``scripts/video_sample.py:446-448``,
RetinaFace landmarks + ``cv2.estimateAffinePartial2D``); that detector is not part of this package (network weights
and cv2 unavailable), so ``get_crop_face`` delegates to a ``detector`` object when one is supplied (e.g. the reference's
own helper) and raises otherwise.

OpenCV's arithmetic is restated from its published algorithm (opencv 4.4: fixed-point source coordinates, a = -0.75
cubic weight table, constant-border rule, separable float64 Gaussian with reflect-101): **parity unpinned** -- cv2 is not
installable in the build container and the reference holds no fixture of these calls.
"""
import numpy as np
import torch

from .. import ops

# facelib/utils/face_restoration_helper.py:283-303
MASK_COLORMAP = [0, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 0, 0, 0, 0, 0]


def invert_affine(M):
    """``cv2.invertAffineTransform`` (imgwarp.cpp), in double on the host: six numbers per face."""
    M = np.asarray(M, dtype=np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = M[1, 1] * D, M[0, 0] * D, -M[0, 1] * D, -M[1, 0] * D
    return np.array([[A11, A12, -A11 * M[0, 2] - A12 * M[1, 2]],
                     [A21, A22, -A21 * M[0, 2] - A22 * M[1, 2]]], dtype=np.float64)


def gaussian_kernel(ksize, sigma):
    """``cv2.getGaussianKernel(ksize, sigma, CV_64F)``."""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp((-0.5 / (sigma * sigma)) * x * x)
    return k * (1.0 / k.sum())


class FaceRestoreHelper(object):
    """``FaceRestoreHelper(face_size=512, device=..., face_parse=ParseNet)``: the crop / paste half of the reference class."""

    def __init__(self, face_size=512, crop_ratio=(1, 1), det_model="retinaface_resnet50", save_ext="png",
                 template_3points=False, device=None, face_parse=None, detector=None):
        assert crop_ratio[0] >= 1 and crop_ratio[1] >= 1, "crop ration only supports >=1"
        self.face_size = (int(face_size * crop_ratio[1]), int(face_size * crop_ratio[0]))    # (w, h), as the reference
        self.device = torch.device(device if device is not None else "cuda")
        self.face_parse = face_parse          # flair_amd.guided_diffusion.parsenet.ParseNet (needed by inverse_faces)
        self.detector = detector
        self._const = {}
        self._minv_cache = {}

    # -- detection half (not built: RetinaFace weights are a network download, alignment is cv2 glue)
    def get_crop_face(self, *args, **kwargs):
        if self.detector is not None:
            return self.detector.get_crop_face(*args, **kwargs)
        raise NotImplementedError("flair_amd: face detection / landmark alignment (RetinaFace, "
                                  "face_restoration_helper.py:120-223) is not part of this package; pass the affine "
                                  "matrices in, or construct the helper with detector=<the reference's FaceRestoreHelper>")

    def _consts(self, dev):
        c = self._const.get(dev)
        if c is None:
            c = (torch.tensor(MASK_COLORMAP, dtype=torch.float64, device=dev),
                 torch.from_numpy(gaussian_kernel(101, 26.0)).to(dev))
            self._const[dev] = c
        return c

    def _minv(self, mats, dev, twice=False):
        """The dst -> src matrices warpAffine derives from its argument (its inverse, in double): (N, 6) on the GPU.
        ``twice``: of the INVERSE matrices (inverse_faces warps with inverse_affine, which warpAffine inverts again).
        Cached per matrix list by content: the sampler passes the same list to three calls in every denoising step, and an
        upload from pageable host memory is a synchronous copy in the middle of the step."""
        arr = np.stack([np.asarray(m, dtype=np.float64).reshape(2, 3) for m in mats])
        key = (arr.tobytes(), str(dev), bool(twice))
        hit = self._minv_cache.get(key)
        if hit is None:
            src = [invert_affine(m) for m in arr] if twice else list(arr)
            hit = torch.from_numpy(np.stack([invert_affine(m).reshape(6) for m in src])).to(dev)
            if len(self._minv_cache) >= 64:
                self._minv_cache.clear()
            self._minv_cache[key] = hit
        return hit

    def get_crop_face_from_affine_matrices(self, bathed_imgs, affine_matrices):
        """(B, 3, H, W) in [-1, 1] -> (B, 3, face_h, face_w) in [-1, 1]: per frame
        ``cv2.warpAffine(clamp((x+1)/2, 0, 1)*255, M, face_size, INTER_CUBIC, BORDER_CONSTANT, (135, 133, 132))``,
        ``/ 255``, ``(y - 0.5) / 0.5``, clamp -- one launch for the batch."""
        if len(affine_matrices) == 0:
            return None
        x = bathed_imgs.float().contiguous()
        if len(affine_matrices) != x.shape[0]:
            raise ValueError("one affine matrix per frame is needed")
        w, h = self.face_size
        return ops.warp_affine_cubic(x, self._minv(affine_matrices, x.device), (h, w), border=(135.0, 133.0, 132.0),
                                     pre=True, post=True)

    def get_inverse_affine(self, affine_matrices):
        return [invert_affine(m) for m in affine_matrices]

    def inverse_faces(self, restored_face_imgs, affine_matrices):
        """(B, 3, h, w) restored faces -> (faces warped back (B, 3, h, w) in [-1, 1], paste masks (B, 1, h, w) float32)."""
        if self.face_parse is None:
            raise RuntimeError("FaceRestoreHelper.inverse_faces needs face_parse=ParseNet(...) (facelib/parsing/parsenet.py)")
        x = restored_face_imgs.float().contiguous()
        B, _, h, w = x.shape
        lut, kern = self._consts(x.device)
        net = self.face_parse
        logits = net.out_mask_conv.run(net._features(x))                       # face_parse(x)[0], still NHWC
        _, idx = ops.argmax_codebook(logits, net.parsing_ch, torch.zeros((net.parsing_ch, 1), dtype=torch.float32,
                                                                          device=x.device))
        mask = ops.face_mask_blur(idx, B, h, w, lut, kern, repeats=2, edge=10, div=255.0)
        # warpAffine(., inverse_affine) inverts its argument again: dst -> src = inv(inv(M)), rounded as OpenCV rounds it
        minv = self._minv(affine_matrices, x.device, twice=True)
        inv_faces = ops.warp_affine_cubic(x, minv, (h, w), pre=True, post=True)
        inv_masks = ops.warp_affine_cubic(mask, minv, (h, w))
        return inv_faces, inv_masks
