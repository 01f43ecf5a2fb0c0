"""CPU restatement of FLAIR's degradation operators (restore_fn side of the sampler).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, with plain numpy / torch CPU:
  * guided_diffusion/pseudoSR.py + imresize_pseudoSR.py  -> BlurOperator
  * guided_diffusion/jpeg.py + dct.py                    -> jpeg_encode / jpeg_decode
  * guided_diffusion/restore_util.py (SRConv)            -> SeparableSR
  * guided_diffusion/resizer.py                          -> resizer_contributions / resize_apply
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import convolve2d


# ------------------------------------------------------------------ blur x f operator
def _strides(shape, factor, align_center=False):
    """imresize_pseudoSR.py:81-94."""
    f = int(max(factor, 1 / factor))
    if not align_center:
        post = int(np.floor(f / 2))
        return np.array([f - post - 1] * 2), np.array([post] * 2)
    half = np.ceil(np.array(shape[:2]) / 2 * (factor if factor > 1 else 1))
    pre = np.mod(half, f)
    pre[pre == 0] = f
    pre = (pre - 1).astype(int)
    return pre, f - pre - 1


def _center_of_mass_pad(k, f):
    """imresize_pseudoSR.py:121-157 (Center_Mass)."""
    n = k.shape[0]
    gx, gy = np.meshgrid(np.arange(n), np.arange(n))
    mx = convolve2d(gx, k, mode="valid") + 1
    my = convolve2d(gy, k, mode="valid") + 1
    xp, yp = 2 * (n / 2 - mx), 2 * (n / 2 - my)
    d = np.round(np.abs(yp)) - np.round(np.abs(xp))
    pads = {"x": [np.maximum(0, -xp), np.maximum(0, xp)], "y": [np.maximum(0, -yp), np.maximum(0, yp)]}

    def rnd(v):
        return int(np.round(np.asarray(v, dtype=np.float64).reshape(-1)[0]))

    def widen(axis, extra):
        pre, post = pads[axis]
        right = np.round(post) - post - (np.round(pre) - pre)
        pre, post = rnd(pre), rnd(post)
        lo, hi = int(np.floor(extra / 2)), int(np.ceil(extra / 2))
        pads[axis] = [pre + lo, post + hi] if right > 0 else [pre + hi, post + lo]

    if d > 0:
        pads["y"] = [rnd(v) for v in pads["y"]]
        widen("x", d)
    elif d < 0:
        pads["x"] = [rnd(v) for v in pads["x"]]
        widen("y", -d)
    k = np.pad(k, ((rnd(pads["y"][0]), rnd(pads["y"][1])), (rnd(pads["x"][0]), rnd(pads["x"][1]))))
    tot = np.sqrt(np.sum(k ** 2))
    energy = np.array([1.0] + [np.sqrt(np.sum(k[j:-j, j:-j] ** 2)) / tot
                               for j in range(1, int(np.ceil(k.shape[0] / 2)))])
    cut = np.argwhere(energy < 0.99)[0][0] * np.ones(2, dtype=int)
    side = 0
    while (k.shape[0] - cut.sum() - 1 + (f + 1) % 2) % f != 0:
        cut[side] -= 1
        side ^= 1
    k = k[cut[0]:-cut[1], cut[0]:-cut[1]]
    return k / k.sum()


def blur_filters(kernel, f, lower_magnitude_bound=0.01, nfft_add=36):
    """ds_kernel and inv_hTh of pseudoSR.py:47-171 for an explicit blur kernel (kernel_indx>=8)."""
    pre, post = _strides(None, f)
    k = _center_of_mass_pad(np.asarray(kernel, dtype=np.float64), f) * f ** 2
    k = np.pad(k, ((max(0, post[0] - pre[0]), max(0, pre[0] - post[0])),
                   (max(0, post[1] - pre[1]), max(0, pre[1] - post[1]))))
    ds = np.rot90(k, 2).astype(np.float32).astype(np.float64) / f ** 2
    hth = convolve2d(ds, np.rot90(ds, 2)) * f ** 2
    p0, _ = _strides(hth.shape, 1 / f, align_center=True)
    hth = hth[p0[0]::f, p0[1]::f]
    h = nfft_add // 2
    spec = np.fft.fft2(np.pad(hth, h))
    spec = spec * np.maximum(1, lower_magnitude_bound / np.abs(spec))
    inv = np.real(np.fft.ifft2(1 / spec))
    r, c = divmod(int(np.argmax(inv)), inv.shape[0])
    if not (math.ceil(inv.shape[0] / 2) == r - 1 and math.ceil(inv.shape[1] / 2) == c - 1):
        m = min(inv.shape[0] - r - 1, inv.shape[0] - c - 1, r, c)
        inv = inv[r - m:r + m + 1, c - m:c + m + 1]
    extra = inv.shape[0] // 2 - 26
    if extra > 0:
        inv = inv[extra:-extra, extra:-extra]
    return ds, inv, pre, post


class BlurOperator:
    """pseudoSR_PyTorch (pseudoSR.py:174-281) with torch CPU convolutions."""

    def __init__(self, kernel, f=4):
        self.f = f
        self.ds_kernel, self.inv_hTh, self.pre, self.post = blur_filters(kernel, f)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()[None, None].repeat(3, 1, 1, 1)  # noqa: E731
        self.w_inv = t(self.inv_hTh)
        self.w_down = t(np.rot90(self.ds_kernel, 2))
        self.w_up = t(self.ds_kernel * f ** 2)

    @staticmethod
    def _filt(x, w):
        p = w.shape[-1] // 2
        return F.conv2d(F.pad(x, (p, p, p, p), mode="replicate"), w, groups=3)

    def down(self, x):
        y = self._filt(x, self.w_down)
        return y[:, :, self.pre[0]::self.f, self.pre[1]::self.f]

    def inv(self, x):
        return self._filt(x, self.w_inv)

    def up(self, x):
        n, c, h, w = x.shape
        z = x.new_zeros(n, c, h * self.f, w * self.f)
        z[:, :, self.pre[0]::self.f, self.pre[1]::self.f] = x
        return self._filt(z, self.w_up)

    def a_forward(self, x):
        """pseudoSR_PyTorch.A with its default arguments (pseudoSR.py:283-295 ->
        imresize_efficient, imresize_pseudoSR.py:163-178): reflect-pad, correlate with
        rot180(ds_kernel), keep [pre::1] (scale_factor=1.0 -> no decimation)."""
        p = self.w_down.shape[-1] // 2
        y = F.conv2d(F.pad(x, (p, p, p, p), mode="reflect"), self.w_down, groups=3)
        return y[:, :, self.pre[0]:, self.pre[1]:]

    def a_pinv(self, lr, x=None, codec=None):
        lr = lr[:, -3:]
        if x is None:
            return self.up(self.inv(lr))
        d = self.down(x)
        if codec is not None:
            d = codec(d)
        return self.up(self.inv(d)) - self.up(self.inv(lr))


# ------------------------------------------------------------------------------ JPEG
_Q_LUMA = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
           14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
           49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]
_Q_CHROMA = [17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
             47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32


def quant_tables(qf):
    """jpeg.py:35-65."""
    s = (5000 / qf) if qf < 50 else (200 - 2 * qf)
    out = []
    for base in (_Q_LUMA, _Q_CHROMA):
        q = torch.floor((s * torch.tensor(base) + 50) / 100)
        q[q <= 0] = 1
        q[q > 255] = 255
        out.append(q.view(8, 8))
    return out


def _dct_matrix():
    """The matrix LinearDCT(8,'dct','ortho') applies (dct.py:31-63,167-190): built, like the
    reference, by running the FFT-based DCT-II on the identity in f32."""
    n = 8
    x = torch.eye(n)
    v = torch.cat([x[:, ::2], x[:, 1::2].flip([1])], dim=1)
    vc = torch.view_as_real(torch.fft.fft(v, dim=1))
    k = -torch.arange(n, dtype=x.dtype)[None, :] * np.pi / (2 * n)
    V = vc[:, :, 0] * torch.cos(k) - vc[:, :, 1] * torch.sin(k)
    V[:, 0] /= np.sqrt(n) * 2
    V[:, 1:] /= np.sqrt(n / 2) * 2
    return (2 * V).t()          # weight of the linear layer: y = x @ W^T


def _idct_matrix():
    """LinearDCT(8,'idct','ortho') (dct.py:66-104)."""
    n = 8
    X = torch.eye(n)
    xv = X / 2
    xv = xv.clone()
    xv[:, 0] *= np.sqrt(n) * 2
    xv[:, 1:] *= np.sqrt(n / 2) * 2
    k = torch.arange(n, dtype=X.dtype)[None, :] * np.pi / (2 * n)
    wr, wi = torch.cos(k), torch.sin(k)
    vtr = xv
    vti = torch.cat([xv[:, :1] * 0, -xv.flip([1])[:, :-1]], dim=1)
    vr = vtr * wr - vti * wi
    vi = vtr * wi + vti * wr
    v = torch.fft.irfft(torch.view_as_complex(torch.stack([vr, vi], dim=2).contiguous()), n=n, dim=1)
    x = v.new_zeros(v.shape)
    x[:, ::2] += v[:, : n - n // 2]
    x[:, 1::2] += v.flip([1])[:, : n // 2]
    return x.t()


def _lin2d(x, w):
    """apply_linear_2d (dct.py:193-202)."""
    y = F.linear(x, w)
    return F.linear(y.transpose(-1, -2), w).transpose(-1, -2)


def _blocks(x):          # (n,c,s,s) -> (n*blocks, c, 8, 8) in unfold order
    n, c, s, _ = x.shape
    u = F.unfold(x, kernel_size=8, stride=8).transpose(2, 1)          # n, L, c*64
    return u.reshape(-1, c, 8, 8)


def _unblocks(b, n, c, s):
    u = b.reshape(n, (s // 8) ** 2, 64 * c).transpose(2, 1)
    return F.fold(u, output_size=(s, s), kernel_size=8, stride=8)


def jpeg_encode(x, qf):
    """jpeg.py:72-114 -> [quantised luma (n,1,s,s), quantised chroma (n,2,s/2,s/2)]."""
    x = (x + 1) / 2 * 255
    n, _, s, _ = x.shape
    m = torch.tensor([[0.299, 0.587, 0.114], [-0.1687, -0.3313, 0.5], [0.5, -0.4187, -0.0813]])
    ycc = torch.einsum("nchw,kc->nkhw", x, m).clone()
    ycc[:, 1:] += 128
    luma, chroma = ycc[:, 0:1], ycc[:, 1:, ::2, ::2]
    q1, q2 = quant_tables(qf)
    D = _dct_matrix()
    bl = _lin2d(_blocks(luma).reshape(-1, 8, 8) - 128, D).view(-1, 1, 8, 8)
    bc = _lin2d(_blocks(chroma).reshape(-1, 8, 8) - 128, D).view(-1, 2, 8, 8)
    bl = (bl / q1).round()
    bc = (bc / q2).round()
    return [_unblocks(bl, n, 1, s), _unblocks(bc, n, 2, s // 2)]


def jpeg_decode(code, qf):
    """jpeg.py:117-167."""
    luma, chroma = code
    n, _, s, _ = luma.shape
    q1, q2 = quant_tables(qf)
    Di = _idct_matrix()
    bl = _lin2d((_blocks(luma) * q1).reshape(-1, 8, 8), Di) + 128
    bc = _lin2d((_blocks(chroma) * q2).reshape(-1, 8, 8), Di) + 128
    luma = _unblocks(bl.reshape(-1, 1, 8, 8), n, 1, s)
    chroma = _unblocks(bc.reshape(-1, 2, 8, 8), n, 2, s // 2)
    chroma = chroma.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    ycc = torch.cat([luma, chroma], dim=1).clone()
    ycc[:, 1:] -= 128
    m = torch.tensor([[1.0, -3.68199903e-05, 1.40198758], [1.0, -3.44113281e-01, -7.14103821e-01],
                      [1.0, 1.77197812, -1.34583413e-04]])
    rgb = torch.einsum("nchw,kc->nkhw", ycc, m)
    return rgb / 255 * 2 - 1


# ---------------------------------------------------------------- separable SR (SRConv)
def bicubic_taps(factor, a=-0.5):
    """The 4*factor-tap bicubic kernel of scripts/video_sample.py:205-226."""
    def kfun(x):
        x = abs(x)
        if x <= 1:
            return (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1
        if x < 2:
            return a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a
        return 0.0
    n = factor * 4
    k = np.array([kfun((1 / factor) * (i - np.floor(n / 2) + 0.5)) for i in range(n)])
    return k / k.sum()


class SeparableSR:
    """SRConv (restore_util.py:102-227): 1-D strided conv matrix with reflect padding, its SVD
    (singular values < 3e-2 zeroed) and the Kronecker-structured A / A_pinv on (n, 3*S*S)."""

    def __init__(self, kernel, channels, img_dim, stride):
        self.S, self.c, self.f = img_dim, channels, stride
        small = img_dim // stride
        self.s = small
        A = torch.zeros(small, img_dim)
        half = kernel.shape[0] // 2
        for i in range(stride // 2, img_dim + stride // 2, stride):
            for j in range(i - half, i + half):
                je = -j - 1 if j < 0 else ((img_dim - 1) - (j - img_dim) if j >= img_dim else j)
                A[i // stride, je] += float(kernel[j - i + half])
        self.A_small = A
        U, sv, V = torch.svd(A, some=False)
        sv = sv.clone()
        sv[sv < 3e-2] = 0
        self.U, self.sv, self.V = U, sv, V

    def A(self, x):
        """x: (n, c*S*S) -> (n, c*s*s): A_small X A_small^T with the thresholded spectrum."""
        n = x.shape[0]
        X = x.reshape(n * self.c, self.S, self.S)
        Asm = self.U @ torch.diag(self.sv) @ self.V[:, : self.s].t()
        return (Asm @ X @ Asm.t()).reshape(n, -1)

    def A_pinv(self, y):
        n = y.shape[0]
        Y = y.reshape(n * self.c, self.s, self.s)
        inv = torch.where(self.sv > 0, 1.0 / self.sv, torch.zeros_like(self.sv))
        P = self.V[:, : self.s] @ torch.diag(inv) @ self.U.t()
        # note: the 2-D singular values are products s_i*s_j; zeroed entries stay zero
        return (P @ Y @ P.t()).reshape(n, -1)


# --------------------------------------------------------------------------- Resizer
def _cubic(x):
    ax = np.abs(x)
    return ((1.5 * ax ** 3 - 2.5 * ax ** 2 + 1) * (ax <= 1)
            + (-0.5 * ax ** 3 + 2.5 * ax ** 2 - 4 * ax + 2) * ((1 < ax) & (ax <= 2)))


def resizer_contributions(in_len, out_len, scale, antialiasing=True, kernel_width=4.0):
    """resizer.py:103-166 for the cubic kernel: weights (out, taps) and reflected indices."""
    aa = antialiasing and scale < 1
    kern = (lambda a: scale * _cubic(scale * a)) if aa else _cubic
    kw = kernel_width / scale if aa else kernel_width
    out_c = np.arange(1, out_len + 1) - (out_len - in_len * scale) / 2
    match = out_c / scale + 0.5 * (1 - 1 / scale)
    left = np.floor(match - kw / 2)
    width = int(np.ceil(kw) + 2)
    fov = (left[:, None] + np.arange(width) - 1).astype(np.int16).astype(np.int64)
    w = kern(match[:, None] - fov - 1.0)
    sw = w.sum(axis=1)
    sw[sw == 0] = 1.0
    w = w / sw[:, None]
    mirror = np.concatenate([np.arange(in_len), np.arange(in_len - 1, -1, -1)])
    fov = mirror[np.mod(fov, mirror.shape[0])]
    keep = np.any(w, axis=0)
    return w[:, keep], fov[:, keep]


def resize_apply(x, scale, antialiasing=True):
    """Resizer.forward (resizer.py:54-73) on the last two dims of (n,c,h,w), both by `scale`."""
    for dim in (2, 3):
        n_in = x.shape[dim]
        n_out = int(np.ceil(n_in * scale))
        w, fov = resizer_contributions(n_in, n_out, scale, antialiasing)
        w = torch.tensor(w, dtype=torch.float32)
        idx = torch.tensor(fov, dtype=torch.long)
        xm = x.movedim(dim, -1)
        y = (xm[..., idx] * w).sum(-1)
        x = y.movedim(-1, dim)
    return x
