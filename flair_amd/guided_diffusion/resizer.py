"""MATLAB-style ``Resizer`` on MI355X (reference: guided_diffusion/resizer.py:7-197; named by the
north-star although nothing in the reference calls it).  The per-dimension contribution tables
(``contributions``, resizer.py:103-166) are host numpy constants; applying them is a gather +
weighted sum per axis in ``flair_gather_mac_f32``.  Cubic / linear / box / lanczos kernels.
"""
from math import pi

import numpy as np
import torch

from .. import ops


def cubic(x):
    ax = np.abs(x)
    return ((1.5 * ax ** 3 - 2.5 * ax ** 2 + 1) * (ax <= 1)
            + (-0.5 * ax ** 3 + 2.5 * ax ** 2 - 4 * ax + 2) * ((1 < ax) & (ax <= 2)))


def lanczos2(x):
    e = np.finfo(np.float32).eps
    return ((np.sin(pi * x) * np.sin(pi * x / 2) + e) / ((pi ** 2 * x ** 2 / 2) + e)) * (abs(x) < 2)


def lanczos3(x):
    e = np.finfo(np.float32).eps
    return ((np.sin(pi * x) * np.sin(pi * x / 3) + e) / ((pi ** 2 * x ** 2 / 3) + e)) * (abs(x) < 3)


def box(x):
    return ((-0.5 <= x) & (x < 0.5)) * 1.0


def linear(x):
    return (x + 1) * ((-1 <= x) & (x < 0)) + (1 - x) * ((0 <= x) & (x <= 1))


_KERNELS = {"cubic": (cubic, 4.0), "lanczos2": (lanczos2, 4.0), "lanczos3": (lanczos3, 6.0),
            "box": (box, 1.0), "linear": (linear, 2.0), None: (cubic, 4.0)}


class Resizer(torch.nn.Module):
    def __init__(self, in_shape, scale_factor=None, output_shape=None, kernel=None, antialiasing=True):
        super().__init__()
        scale, out_shape = self._fix(in_shape, output_shape, scale_factor)
        method, width = _KERNELS[kernel]
        antialiasing = bool(antialiasing) and bool(np.any(np.array(scale) < 1))
        order = np.argsort(np.array(scale))
        self.sorted_dims = [int(d) for d in order if scale[d] != 1]
        self._tables = []
        for d in self.sorted_dims:
            w, fov = self.contributions(in_shape[d], int(out_shape[d]), scale[d], method, width, antialiasing)
            # stored transposed: (taps, out_len)
            self._tables.append((torch.tensor(np.ascontiguousarray(fov.T), dtype=torch.int32),
                                 torch.tensor(np.ascontiguousarray(w.T), dtype=torch.float32)))
        self._dev = {}

    @staticmethod
    def _fix(in_shape, out_shape, scale):
        if scale is not None:
            if np.isscalar(scale) and len(in_shape) > 1:
                scale = [scale, scale]
            scale = list(scale)
            scale = [1] * (len(in_shape) - len(scale)) + scale
        if out_shape is not None:
            out_shape = list(in_shape[len(out_shape):]) + list(np.uint(np.array(out_shape)))
        if scale is None:
            scale = 1.0 * np.array(out_shape) / np.array(in_shape)
        if out_shape is None:
            out_shape = np.uint(np.ceil(np.array(in_shape) * np.array(scale)))
        return scale, out_shape

    @staticmethod
    def contributions(in_length, out_length, scale, kernel, kernel_width, antialiasing):
        fixed = (lambda a: scale * kernel(scale * a)) if antialiasing else kernel
        kernel_width = kernel_width / scale if antialiasing else kernel_width
        out_c = np.arange(1, out_length + 1) - (out_length - in_length * scale) / 2
        match = out_c / scale + 0.5 * (1 - 1 / scale)
        left = np.floor(match - kernel_width / 2)
        span = int(np.ceil(kernel_width) + 2)
        fov = np.squeeze(np.int16(np.expand_dims(left, axis=1) + np.arange(span) - 1))
        w = fixed(1.0 * np.expand_dims(match, axis=1) - fov - 1)
        sw = np.sum(w, axis=1)
        sw[sw == 0] = 1.0
        w = 1.0 * w / np.expand_dims(sw, axis=1)
        mirror = np.uint(np.concatenate((np.arange(in_length), np.arange(in_length - 1, -1, step=-1))))
        fov = mirror[np.mod(fov, mirror.shape[0])]
        keep = np.nonzero(np.any(w, axis=0))
        return np.squeeze(w[:, keep]), np.squeeze(fov[:, keep])

    def forward(self, in_tensor):
        if not in_tensor.is_cuda:
            raise RuntimeError("flair_amd.Resizer runs on the MI355X only (no CPU path)")
        x = in_tensor.float().contiguous()
        key = x.device
        if key not in self._dev:
            self._dev[key] = [(f.to(key), w.to(key)) for f, w in self._tables]
        for dim, (fov, w) in zip(self.sorted_dims, self._dev[key]):
            shape = list(x.shape)
            outer = int(np.prod(shape[:dim])) if dim > 0 else 1
            inner = int(np.prod(shape[dim + 1:])) if dim + 1 < len(shape) else 1
            y = ops.gather_mac(x, outer, shape[dim], inner, fov, w)
            shape[dim] = fov.shape[1]
            x = y.reshape(shape)
        return x
