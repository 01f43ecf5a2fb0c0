"""Gaussian-blur x4 degradation operator of the gaussian / jpeg tasks on MI355X.

Mirrors the interface of the reference's ``guided_diffusion/pseudoSR.py`` +
``imresize_pseudoSR.py`` that ``scripts/video_sample.py:248-261,183-193`` uses:
``Get_pseudoSR_Conf(sf)``, ``pseudoSR(conf, upscale_kernel=..., kernel_indx=...)``,
``.WrapArchitecture_PyTorch().to(device)`` and the resulting operator's
``A_pinv(LR, generated_image, jpeg_decode=, jpeg_encode=)`` / ``A(HR)``.

The two filters (anti-aliasing kernel ``ds_kernel`` and the regularised inverse of
h^T h, ``inv_hTh``) are built once on the host in float64/float32 numpy exactly as
the reference does (they are 9x9 and 39x39 constants of the operator).  Applying
them is the per-step work and runs in ``flair_depthwise_filter`` on the GPU; the
operator is linear, so ``A_pinv`` evaluates
``Up(InvHtH(codec(Down(x)) - LR))`` in three launches instead of the reference's
six filter passes (pseudoSR.py:263-276) -- identical up to f32 rounding.
Unlike the reference, nothing is written to the working directory (the reference
dumps ``rot59.mat`` as a side effect, imresize_pseudoSR.py:59).
"""
import numpy as np
import torch
from scipy.signal import convolve2d

from .. import ops


def Get_pseudoSR_Conf(sf):
    """pseudoSR.py:383-396."""
    class conf:
        scale_factor = sf
        avoid_skip_connections = False
        generate_HR_image = False
        pseudo_pseudoSR_supplement = False
        desired_inv_hTh_energy_portion = 1 - 1e-6
        filter_pertubation_limit = 1.1
        sigmoid_range_limit = False
        lower_magnitude_bound = 0.01

    return conf


def calc_strides(array, factor, align_center=False):
    """imresize_pseudoSR.py:81-94: how the f-1 padding samples of a stride-f resampler are
    split before/after each kept sample."""
    f = int(max(factor, 1 / factor))
    if align_center:
        half = np.ceil(np.array(array.shape[:2]) / 2 * (factor if factor > 1 else 1))
        pre = np.mod(half, f)
        pre[pre == 0] = f
        pre = (pre - 1).astype(np.int32)
        return pre, f - pre - 1
    post = (np.floor(f / 2) * np.ones(2)).astype(np.int32)
    return (f - post - 1).astype(np.int32), post


def _energy_profile(filt):
    """sqrt-energy kept when peeling n frames off a square filter, relative to all of it
    (imresize_pseudoSR.py:159-161)."""
    n = int(np.ceil(filt.shape[0] / 2))
    e = [np.sqrt(np.sum(filt ** 2))] + [np.sqrt(np.sum(filt[k:-k, k:-k] ** 2)) for k in range(1, n)]
    return np.array(e) / e[0]


def _rint(v):
    return int(np.round(np.asarray(v, dtype=np.float64).reshape(-1)[0]))


def center_mass(kernel, ds_factor):
    """Pad a blur kernel so that its centre of mass sits in the middle, then trim margins
    holding <1 % of the energy while keeping (size - 1 + [f even]) divisible by f
    (imresize_pseudoSR.py:121-157)."""
    n = kernel.shape[0]
    assert kernel.shape[0] == kernel.shape[1]
    xs, ys = np.meshgrid(np.arange(n), np.arange(n))
    cx = convolve2d(xs, kernel, mode="valid") + 1
    cy = convolve2d(ys, kernel, mode="valid") + 1
    x_pad, y_pad = 2 * (n / 2 - cx), 2 * (n / 2 - cy)
    diff = np.round(np.abs(y_pad)) - np.round(np.abs(x_pad))
    pre_x, post_x = np.maximum(0, -x_pad), np.maximum(0, x_pad)
    pre_y, post_y = np.maximum(0, -y_pad), np.maximum(0, y_pad)

    def spread(pre, post, extra):
        lean_right = np.round(post) - post - (np.round(pre) - pre)
        pre, post = _rint(pre), _rint(post)
        if lean_right > 0:
            return pre + int(np.floor(extra / 2)), post + int(np.ceil(extra / 2))
        return pre + int(np.ceil(extra / 2)), post + int(np.floor(extra / 2))

    if diff > 0:
        pre_y, post_y = _rint(pre_y), _rint(post_y)
        pre_x, post_x = spread(pre_x, post_x, diff)
    elif diff < 0:
        pre_x, post_x = _rint(pre_x), _rint(post_x)
        pre_y, post_y = spread(pre_y, post_y, -diff)
    k = np.pad(kernel, ((_rint(pre_y), _rint(post_y)), (_rint(pre_x), _rint(post_x))), mode="constant")
    assert k.shape[0] == k.shape[1]
    drop = np.argwhere(_energy_profile(k) < 0.99)[0][0] * np.ones(2, dtype=np.int32)
    which = 0
    while np.mod(k.shape[0] - np.sum(drop) - 1 + np.mod(ds_factor + 1, 2), ds_factor) != 0:
        drop[which] -= 1
        which = (which + 1) % 2
    k = k[drop[0]:-drop[1], drop[0]:-drop[1]]
    return k / np.sum(k)


def antialiasing_kernel(upscale_kernel, ds_factor):
    """``Return_kernel`` (pseudoSR.py:352-364) for a supplied blur kernel with kernel_indx >= 8
    (imresize_pseudoSR.py:25-37,56): centred kernel * f^2, padded for the uneven stride split,
    rotated by 180 degrees and divided by f^2."""
    assert abs(1 - np.sum(upscale_kernel)) < np.finfo(np.float32).eps, "kernel must sum to 1"
    pre, post = calc_strides(None, ds_factor)
    pad_after = np.maximum(0, pre - post)
    pad_before = np.maximum(0, post - pre)
    k = center_mass(upscale_kernel, ds_factor) * ds_factor ** 2
    assert np.all(np.mod(np.array(k.shape) + pad_after + pad_before - 1, ds_factor) == 0)
    k = np.pad(k, ((pad_before[0], pad_after[0]), (pad_before[1], pad_after[1])), mode="constant")
    # values rounded to f32, carried in float64 (the reference divides by an int32 array scalar)
    return np.rot90(k, 2).astype(np.float32).astype(np.float64) / ds_factor ** 2, pre, post


def inverse_hTh(ds_kernel, ds_factor, lower_magnitude_bound, nfft_add=36):
    """pseudoSR.py:123-171: h^T h sampled on the LR grid, inverted in the Fourier domain with
    its magnitude bounded from below, re-centred on its peak and cropped to +-26 taps."""
    ds_kernel = np.asarray(ds_kernel, dtype=np.float64)   # the FFT must not run in complex64
    hTh = convolve2d(ds_kernel, np.rot90(ds_kernel, 2)) * ds_factor ** 2
    pre, _ = calc_strides(hTh, 1 / ds_factor, align_center=True)
    hTh = hTh[pre[0]::ds_factor, pre[1]::ds_factor]
    p = int(nfft_add / 2)
    spec = np.fft.fft2(np.pad(hTh, ((p, p), (p, p)), mode="constant"))
    spec = spec * np.maximum(1, lower_magnitude_bound / np.abs(spec))
    inv = np.real(np.fft.ifft2(1 / spec))
    r, c = np.argmax(inv) // inv.shape[0], np.mod(np.argmax(inv), inv.shape[0])
    if not np.all(np.equal(np.ceil(np.array(inv.shape) / 2), np.array([r, c]) - 1)):
        half = np.min([inv.shape[0] - r - 1, inv.shape[0] - c - 1, r, c])
        inv = inv[r - half:r + half + 1, c - half:c + half + 1]
    drop = inv.shape[0] // 2 - 26
    if drop > 0:
        inv = inv[drop:-drop, drop:-drop]
    return inv


class pseudoSR:
    """pseudoSR.py:47-171 (the parts ``video_sample.py`` needs)."""

    NFFT_add = 36

    def __init__(self, conf, upscale_kernel=None, kernel_indx=0):
        if not isinstance(upscale_kernel, np.ndarray) or kernel_indx < 8:
            raise NotImplementedError("flair_amd: pseudoSR needs an explicit blur kernel with "
                                      "kernel_indx >= 8 (what scripts/video_sample.py passes)")
        self.conf = conf
        self.ds_factor = int(np.array(conf.scale_factor, dtype=np.int32))
        self.ds_kernel, self.pre_stride, self.post_stride = antialiasing_kernel(
            np.asarray(upscale_kernel, dtype=np.float64), self.ds_factor)
        self.inv_hTh = inverse_hTh(self.ds_kernel, self.ds_factor, conf.lower_magnitude_bound, self.NFFT_add)
        self.inv_hTh_invalidity_half_size = 26

    def WrapArchitecture_PyTorch(self, grayscale=False):
        return pseudoSR_PyTorch(self, grayscale=grayscale)


class pseudoSR_PyTorch:
    """The operator object (reference: an nn.Module of three Filter_Layers, pseudoSR.py:174-295)."""

    def __init__(self, op, grayscale=False):
        self.ds_factor = op.ds_factor
        self.conf = op.conf
        self.ds_kernel = op.ds_kernel
        self.pre_stride, self.post_stride = op.pre_stride, op.post_stride
        f = self.ds_factor
        # filters as the reference's three Conv2d weights (cross-correlation kernels)
        self._host = dict(inv=np.ascontiguousarray(op.inv_hTh, dtype=np.float32),
                          down=np.ascontiguousarray(np.rot90(op.ds_kernel, 2), dtype=np.float32),
                          up=np.ascontiguousarray(op.ds_kernel * f ** 2, dtype=np.float32))
        self._dev = None
        self.device = None

    def to(self, device):
        self.device = torch.device(device)
        self._dev = {k: torch.from_numpy(v).to(self.device) for k, v in self._host.items()}
        return self

    def eval(self):
        return self

    # ---- the three filter layers ------------------------------------------------------
    def DownscaleOP(self, x):
        k = self._dev["down"]
        f = self.ds_factor
        H, W = x.shape[-2:]
        return ops.depthwise_filter(x, k, pad=k.shape[0] // 2, out_stride=f, out_offset=int(self.pre_stride[0]),
                                    out_hw=(H // f, W // f))

    def Conv_LR_with_Inv_hTh_OP(self, x):
        k = self._dev["inv"]
        return ops.depthwise_filter(x, k, pad=k.shape[0] // 2, out_hw=tuple(x.shape[-2:]))

    def Upscale_OP(self, x):
        k = self._dev["up"]
        f = self.ds_factor
        H, W = x.shape[-2:]
        return ops.depthwise_filter(x, k, pad=k.shape[0] // 2, stuff=f, stuff_offset=int(self.pre_stride[0]),
                                    out_hw=(H * f, W * f))

    # ---- operator ---------------------------------------------------------------------
    def A_pinv(self, LR, generated_image=None, jpeg_decode=None, jpeg_encode=None):
        """pseudoSR.py:248-281: Up(InvHtH(codec(Down(x)))) - Up(InvHtH(LR)) (or Up(InvHtH(LR))
        when no image is given)."""
        if self._dev is None:
            raise RuntimeError("pseudoSR operator must be moved to the GPU with .to(device) first")
        LR = LR[:, -3:].float().contiguous()
        if generated_image is None:
            return self.Upscale_OP(self.Conv_LR_with_Inv_hTh_OP(LR))
        x = generated_image.float().contiguous()
        assert x.shape[-1] % self.ds_factor == 0 and x.shape[-2] % self.ds_factor == 0
        d = self.DownscaleOP(x)
        if jpeg_encode is not None or jpeg_decode is not None:
            enc = jpeg_encode if jpeg_encode is not None else (lambda v: v)
            dec = jpeg_decode if jpeg_decode is not None else (lambda v: v)
            d = dec(enc(d))
        d = ops.axpby(d.contiguous(), LR, 1.0, -1.0)
        return self.Upscale_OP(self.Conv_LR_with_Inv_hTh_OP(d))

    def A(self, HR, scale_factor=1.0, use_zero_padding=False, kk=None):
        """pseudoSR.py:283-295 -> imresize_efficient (imresize_pseudoSR.py:163-178): reflect-pad,
        correlate with rot180(ds_kernel), keep [pre::1/scale]."""
        if scale_factor != 1.0 or use_zero_padding:
            raise NotImplementedError("flair_amd: pseudoSR.A supports the default arguments only")
        k = self._dev["down"]                      # rot180(ds_kernel), the cross-correlation mask
        f = self.ds_factor
        HR = HR.float().contiguous()
        H, W = HR.shape[-2:]
        # NB: with scale_factor=1.0 the reference keeps EVERY filtered sample ([pre::1]); the blur
        # is applied but no decimation happens (imresize_pseudoSR.py:178).
        return ops.depthwise_filter(HR, k, pad=k.shape[0] // 2, out_stride=1, out_offset=0, reflect=True,
                                    out_hw=(H, W))[:, :, int(self.pre_stride[0]):, int(self.pre_stride[1]):]
