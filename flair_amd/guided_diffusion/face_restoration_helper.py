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
                 template_3points=False, device=None, face_parse=None, detector=None, face_det=None):
        assert crop_ratio[0] >= 1 and crop_ratio[1] >= 1, "crop ration only supports >=1"
        self.face_size = (int(face_size * crop_ratio[1]), int(face_size * crop_ratio[0]))    # (w, h), as the reference
        self.device = torch.device(device if device is not None else "cuda")
        self.face_parse = face_parse          # flair_amd.guided_diffusion.parsenet.ParseNet (needed by inverse_faces)
        self.detector = detector              # optional external object with a get_crop_face method (e.g. the reference's helper)
        self.face_det = face_det              # flair_amd.guided_diffusion.retinaface.RetinaFace (built on first use when None)
        self.det_model = det_model
        self.template_3points = template_3points
        self.crop_ratio = crop_ratio
        if template_3points:
            self.face_template = np.array([[192, 240], [319, 240], [257, 371]], dtype=np.float64)
        else:                                 # standard 5 landmarks for FFHQ faces with 512 x 512 (face_restoration_helper.py:90-98)
            self.face_template = np.array([[192.98138, 239.94708], [318.90277, 240.1936], [256.63416, 314.01935],
                                           [201.26117, 371.41043], [313.08905, 371.15118]])
        self.face_template = self.face_template * (face_size / 512.0)
        if crop_ratio[0] > 1:
            self.face_template[:, 1] += face_size * (crop_ratio[0] - 1) / 2
        if crop_ratio[1] > 1:
            self.face_template[:, 0] += face_size * (crop_ratio[1] - 1) / 2
        self._const = {}
        self._minv_cache = {}

    # -- detection half (facelib/utils/face_restoration_helper.py:122-224): RetinaFace on the HIP kernels, landmark alignment on
    # the host (five points per face), the crop itself by flair_warp_affine_cubic
    def _detector(self):
        if self.face_det is None:
            if "retinaface" not in self.det_model:
                raise NotImplementedError(f"flair_amd: det_model={self.det_model!r} is not built (retinaface_resnet50 only)")
            from .retinaface import RetinaFace
            # random initialisation: load detection_Resnet50_Final.pth with face_det.load_state_dict(torch.load(path,
            # weights_only=True)) -- the reference downloads it at construction (facelib/detection/__init__.py:25-48)
            self.face_det = RetinaFace(network_name=self.det_model.split("_", 1)[1], half=False, device=self.device)
        return self.face_det

    def get_crop_face(self, bathed_imgs, only_keep_largest=False, only_center_face=False, resize=None, eye_dist_threshold=None,
                      face_template_resize=None, face_template_x_offset=None, face_template_y_offset=None):
        """(B, 3, H, W) frames in [-1, 1] -> (cropped faces (n, 3, face_h, face_w) in [-1, 1], affine matrices, indices of the
        frames with a face) or (None, None, None).  Detection runs on clamp((x + 1) / 2, 0, 1) * 255 like the reference."""
        if self.detector is not None:
            return self.detector.get_crop_face(bathed_imgs, only_keep_largest, only_center_face, resize, eye_dist_threshold,
                                               face_template_resize, face_template_x_offset, face_template_y_offset)
        from .retinaface_utils import estimate_affine_partial, get_center_face, get_largest_face
        face_template_resize = 1.0 if face_template_resize is None else face_template_resize
        face_template_x_offset = 0.0 if face_template_x_offset is None else face_template_x_offset
        face_template_y_offset = 0.0 if face_template_y_offset is None else face_template_y_offset
        if resize is not None:
            raise NotImplementedError("flair_amd: get_crop_face(resize=...) is not built (the reference multiplies a Python list "
                                      "by the float scale there, face_restoration_helper.py:152, which raises)")
        x = bathed_imgs.float().contiguous().to(self.device)
        B, _, H, W = x.shape
        # VF.normalize(x, [-1] * 3, [2] * 3).clamp(0, 1) * 255 = clamp(127.5 x + 127.5, 0, 255): folded into the detector's
        # mean-subtraction launch (`pre`)
        dets = self._detector().batched_detect_faces(x, 0.5, pre=(127.5, 127.5, 0.0, 255.0))
        # the reference zips the per-frame results with the frames, so a frame WITHOUT detections shifts the pairing
        # (retinaface.py:393-395 skips it); mirrored: results pair with frames in order
        affine_matrices, find_face_idx = [], []
        template = np.stack([self.face_template[:, 0] + face_template_x_offset,
                             self.face_template[:, 1] + face_template_y_offset], axis=1) * face_template_resize
        for idx, bboxes in enumerate(dets):
            landmarks, det_faces = [], []
            for bbox in bboxes:
                eye_dist = np.linalg.norm([bbox[5] - bbox[7], bbox[6] - bbox[8]])
                if eye_dist_threshold is not None and eye_dist < eye_dist_threshold:
                    continue
                n = 11 if self.template_3points else 15
                landmarks.append(np.array([[bbox[i], bbox[i + 1]] for i in range(5, n, 2)]))
                det_faces.append(bbox[0:5])
            if len(det_faces) == 0:
                continue
            if only_keep_largest:
                _, k = get_largest_face(det_faces, H, W)
                landmark = landmarks[k]
            elif only_center_face:
                _, k = get_center_face(det_faces, H, W)
                landmark = landmarks[k]
            else:
                landmark = landmarks[0]
            M = estimate_affine_partial(landmark, template)        # cv2.estimateAffinePartial2D(..., method=cv2.LMEDS)[0]
            affine_matrices.append(M)
            find_face_idx.append(idx)
        if len(affine_matrices) == 0:
            return None, None, None
        cropped = self.get_crop_face_from_affine_matrices(x[find_face_idx].contiguous(), affine_matrices)
        return cropped, affine_matrices, find_face_idx

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
