"""CPU oracle (TEST INFRASTRUCTURE ONLY -- nothing under flair_amd/ may import this) for the un-aligned prior
branch's crop / inverse paste:

    guided_diffusion/facelib/utils/face_restoration_helper.py:225-254  get_crop_face_from_affine_matrices
    guided_diffusion/facelib/utils/face_restoration_helper.py:256-262  get_inverse_affine
    guided_diffusion/facelib/utils/face_restoration_helper.py:264-335  inverse_faces
    guided_diffusion/gaussian_diffusion.py:476-493                     the branch itself

The reference calls OpenCV (opencv-python 4.4.0.46, requirements.txt) for the warps and the blur.  cv2 is not
installable in this container and the reference holds no fixture for these calls: **PARITY UNPINNED**.  What follows
restates OpenCV's published algorithms in numpy, function by function:

  * ``cv2.invertAffineTransform`` and the inversion inside ``cv2.warpAffine`` (imgproc/src/imgwarp.cpp): in double,
    D = 1 / (M00 M11 - M01 M10), A11 = M11 D, A22 = M00 D, A12 = -M01 D, A21 = -M10 D, b = -A [M02, M12];
  * ``cv2.warpAffine(INTER_CUBIC)``: destination pixel (x, y) maps to the fixed-point source position
    X = (cvRound((M01 y + M02) 2^10) + 16 + cvRound(M00 x 2^10)) >> 5 (5 fractional bits; cvRound = round half to even),
    integer part -1 is the first of 4 taps, the fractional part indexes a 32-entry table of cubic weights (a = -0.75)
    evaluated in float; the 16 2-D weights are float products cy[r] * cx[c]; float images accumulate in float, double
    images in double, tap by tap in row-major order; BORDER_CONSTANT: a window completely outside gives the border
    value, a partial window starts from the border value and adds (S - border) * w for the taps inside;
  * ``cv2.getGaussianKernel(101, 26)`` (exp(-x^2 / (2 sigma^2)) normalised in double) and ``cv2.GaussianBlur`` as a
    separable float64 filter with BORDER_REFLECT_101: rows sum_k k[k] S[i + k] in ascending k, columns
    k[c] S[c] + sum_{j >= 1} k[c + j] (S[c + j] + S[c - j]) (filter.simd.hpp RowFilter / SymmColumnFilter).
"""
import numpy as np
import torch

MASK_COLORMAP = [0, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 0, 0, 0, 0, 0]  # frh.py:283-303
BORDER_CROP = (135.0, 133.0, 132.0)                                                                      # frh.py:244


def invert_affine(M):
    """cv2.invertAffineTransform (also what warpAffine does to M when WARP_INVERSE_MAP is not set)."""
    M = np.asarray(M, dtype=np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = M[1, 1] * D, M[0, 0] * D, -M[0, 1] * D, -M[1, 0] * D
    b1 = -A11 * M[0, 2] - A12 * M[1, 2]
    b2 = -A21 * M[0, 2] - A22 * M[1, 2]
    return np.array([[A11, A12, b1], [A21, A22, b2]], dtype=np.float64)


def _cubic_table():
    x = np.arange(32, dtype=np.float32) * np.float32(1.0 / 32)
    A = np.float32(-0.75)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.float32)          # [32][4]


_TAB = _cubic_table()


def warp_affine_cubic(img, M, dsize, border=(0.0, 0.0, 0.0)):
    """cv2.warpAffine(img, M, dsize=(W, H), flags=cv2.INTER_CUBIC, borderMode=cv2.BORDER_CONSTANT, borderValue=border).
    img: (Hs, Ws) or (Hs, Ws, C), float32 or float64; returns the same dtype, shape (H, W[, C])."""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    T = img.dtype.type
    Hs, Ws, C = img.shape
    Wd, Hd = dsize
    Mi = invert_affine(M)
    xs = np.arange(Wd, dtype=np.float64)
    ys = np.arange(Hd, dtype=np.float64)
    adelta = np.rint(Mi[0, 0] * xs * 1024.0).astype(np.int64)
    bdelta = np.rint(Mi[1, 0] * xs * 1024.0).astype(np.int64)
    X0 = np.rint((Mi[0, 1] * ys + Mi[0, 2]) * 1024.0).astype(np.int64) + 16
    Y0 = np.rint((Mi[1, 1] * ys + Mi[1, 2]) * 1024.0).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767) - 1
    sy = np.clip(Y >> 5, -32768, 32767) - 1
    cx = _TAB[X & 31]                                                     # (Hd, Wd, 4) float32
    cy = _TAB[Y & 31]
    out = np.empty((Hd, Wd, C), dtype=img.dtype)
    inner = (sx >= 0) & (sx < max(Ws - 3, 0)) & (sy >= 0) & (sy < max(Hs - 3, 0))
    outside = (sx >= Ws) | (sx + 4 <= 0) | (sy >= Hs) | (sy + 4 <= 0)
    for k in range(C):
        P = img[:, :, k]
        cv = T(border[k] if k < len(border) else 0.0)
        # inner: row sums added to a running sum (remapBicubic's order), in the image's type
        s_in = None
        # border: cv + sum over in-image taps of (S - cv) * w
        s_bd = np.full((Hd, Wd), cv, dtype=img.dtype)
        for r in range(4):
            yi = sy + r
            yok = (yi >= 0) & (yi < Hs)
            yc = np.clip(yi, 0, Hs - 1)
            row = None                      # remapBicubic: a row's four taps are summed first, then added to the running sum
            for c in range(4):
                xi = sx + c
                ok = yok & (xi >= 0) & (xi < Ws)
                S = P[yc, np.clip(xi, 0, Ws - 1)]
                w = (cy[..., r] * cx[..., c]).astype(np.float32).astype(img.dtype)
                term = S * w
                row = term if row is None else row + term
                s_bd = np.where(ok, s_bd + (S - cv) * w, s_bd)
            s_in = row if s_in is None else s_in + row
        res = np.where(inner, s_in, s_bd)
        res = np.where(outside, cv, res)
        out[:, :, k] = res
    return out[:, :, 0] if squeeze else out


def gaussian_kernel(ksize=101, sigma=26.0):
    """cv2.getGaussianKernel(ksize, sigma, CV_64F)."""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp((-0.5 / (sigma * sigma)) * x * x)
    return k * (1.0 / k.sum())


def _reflect101(i, n):
    i = np.asarray(i)
    if n == 1:
        return np.zeros_like(i)
    while True:
        bad = (i < 0) | (i >= n)
        if not bad.any():
            return i
        i = np.where(i < 0, -i, i)
        i = np.where(i >= n, 2 * n - 2 - i, i)


def gaussian_blur(img, ksize=101, sigma=26.0):
    """cv2.GaussianBlur(img (H, W) float64, (ksize, ksize), sigma) with BORDER_REFLECT_101."""
    img = np.asarray(img, dtype=np.float64)
    H, W = img.shape
    k = gaussian_kernel(ksize, sigma)
    r = ksize // 2
    P = img[:, _reflect101(np.arange(-r, W + r), W)]
    acc = k[0] * P[:, 0:W]
    for j in range(1, ksize):
        acc = acc + k[j] * P[:, j:j + W]
    Q = acc[_reflect101(np.arange(-r, H + r), H), :]
    out = k[r] * Q[r:r + H]
    for j in range(1, r + 1):
        out = out + k[r + j] * (Q[r + j:r + j + H] + Q[r - j:r - j + H])
    return out


def get_crop_face_from_affine_matrices(imgs, affine_matrices, face_size=(512, 512)):
    """frh.py:225-254.  imgs: (B, 3, H, W) float32 torch tensor in [-1, 1] -> (B, 3, 512, 512) in [-1, 1]."""
    if len(affine_matrices) == 0:
        return None
    x = ((imgs.float() + 1.0) / 2.0).clamp(0, 1) * 255
    x = x.permute(0, 2, 3, 1).contiguous().numpy()
    crops = [warp_affine_cubic(img, M, face_size, BORDER_CROP).astype(np.float32) for img, M in zip(x, affine_matrices)]
    y = torch.from_numpy(np.stack(crops, axis=0)).permute(0, 3, 1, 2) / 255.0
    return ((y - 0.5) / 0.5).clamp(-1, 1)


def inverse_faces(restored, affine_matrices, parse):
    """frh.py:264-335.  restored: (B, 3, h, w) float32 in [-1, 1]; parse: (B, h, w) integer parsing map
    (face_parse(restored)[0].argmax(1)).  Returns (inv_faces (B, 3, h, w) in [-1, 1], inv_masks (B, 1, h, w) float32)."""
    faces = (((restored.float() + 1.0) / 2.0).clamp(0, 1) * 255).permute(0, 2, 3, 1).contiguous().numpy()
    parse = np.asarray(parse)
    cmap = np.asarray(MASK_COLORMAP, dtype=np.float64)
    inv_faces, inv_masks = [], []
    for face, M, pr in zip(faces, affine_matrices, parse):
        mask = cmap[pr]
        mask = gaussian_blur(mask, 101, 26)
        mask = gaussian_blur(mask, 101, 26)
        thres = 10
        mask[:thres, :] = 0
        mask[-thres:, :] = 0
        mask[:, :thres] = 0
        mask[:, -thres:] = 0
        mask = mask / 255.0
        h, w, _ = face.shape
        inv = invert_affine(M)
        inv_faces.append(warp_affine_cubic(face, inv, (w, h)).astype(np.float32))
        inv_masks.append(warp_affine_cubic(mask, inv, (w, h)).astype(np.float32))
    f = torch.from_numpy(np.stack(inv_faces, axis=0)).permute(0, 3, 1, 2) / 255.0
    f = ((f - 0.5) / 0.5).clamp(-1, 1)
    m = torch.from_numpy(np.stack(inv_masks, axis=0)).unsqueeze(1)
    return f, m


def blend(x0, inv_face, inv_mask):
    """gaussian_diffusion.py:491."""
    return x0 * (1 - inv_mask) + inv_face * inv_mask
