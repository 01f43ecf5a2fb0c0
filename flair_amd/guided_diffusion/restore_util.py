"""Separable super-resolution operator ``SRConv`` of the bicubic tasks on MI355X
(reference: guided_diffusion/restore_util.py:102-227 and the SVD algebra of ``A_functions``
:11-99; constructed by scripts/video_sample.py:205-247, applied by ``bicubic_restore`` :177-181).

The reference builds a 1-D strided-convolution matrix ``A_small`` (reflect padding), takes its
SVD and applies U, S, V^T separably with explicit permutations.  Algebraically
``A(x) = (U S V_s^T) X (U S V_s^T)^T`` and ``A_pinv(y) = (V_s S^+ U^T) Y (V_s S^+ U^T)^T`` per
channel, with the singular values below 3e-2 zeroed.  The two small dense matrices are formed
once on the host (f64 SVD of an (S/f) x S matrix -- a constant of the operator); applying them
is the per-step work and runs as two batched f32 matmuls in ``flair_matmul_f32``.
"""
import numpy as np
import torch

from .. import ops


class SRConv:
    def __init__(self, kernel, channels, img_dim, device, stride=1):
        self.img_dim, self.channels, self.ratio = img_dim, channels, stride
        small = img_dim // stride
        self.y_dim = small
        k = np.asarray(kernel.detach().cpu().numpy() if isinstance(kernel, torch.Tensor) else kernel,
                       dtype=np.float32)
        half = k.shape[0] // 2
        A = np.zeros((small, img_dim), dtype=np.float32)
        for i in range(stride // 2, img_dim + stride // 2, stride):          # restore_util.py:119-131
            for j in range(i - half, i + half):
                je = j
                if je < 0:
                    je = -je - 1
                if je >= img_dim:
                    je = (img_dim - 1) - (je - img_dim)
                A[i // stride, je] += k[j - i + half]
        U, sv, Vt = np.linalg.svd(A.astype(np.float64), full_matrices=True)
        sv = sv.copy()
        sv[sv < 3e-2] = 0                                                     # ZERO threshold, :137-138
        inv = np.where(sv > 0, 1.0 / np.where(sv > 0, sv, 1.0), 0.0)
        Vs = Vt[:small].T                                                     # (S, s)
        fwd = (U * sv) @ Vs.T                                                 # (s, S)
        pinv = (Vs * inv) @ U.T                                               # (S, s)
        self.device = torch.device(device)
        f32 = lambda m: torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to(self.device)  # noqa: E731
        self._fwd, self._fwd_t = f32(fwd), f32(fwd.T)
        self._pinv, self._pinv_t = f32(pinv), f32(pinv.T)
        self.singulars_small = torch.from_numpy(sv.astype(np.float32))

    def _sandwich(self, left, right_t, x, n_in):
        n = x.shape[0]
        X = x.reshape(n * self.channels, n_in, n_in).float().contiguous()
        Y = ops.matmul(left, X)                     # (batch, out, n_in)
        Z = ops.matmul(Y, right_t)                  # (batch, out, out)
        return Z.reshape(n, -1)

    def A(self, vec):
        """(n, c*S*S) -> (n, c*s*s)."""
        return self._sandwich(self._fwd, self._fwd_t, vec, self.img_dim)

    def A_pinv(self, vec):
        """(n, c*s*s) -> (n, c*S*S)."""
        return self._sandwich(self._pinv, self._pinv_t, vec, self.y_dim)

    def singulars(self):
        s = self.singulars_small
        return torch.outer(s, s).reshape(-1).repeat_interleave(3).to(self.device)
