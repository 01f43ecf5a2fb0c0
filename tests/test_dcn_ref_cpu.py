"""The oracle's deform_conv2d (oracle/thirdparty.py, grid_sample based, what every BasicVSR++ fixture and GPU test uses)
against a literal restatement of the reference's OWN DCNv2 source (oracle/dcn_ref.py <- guided_diffusion/dcn/src/
deform_conv_cuda_kernel.cu:468-497,571-633 + deform_conv_cuda.cpp:540-560).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import dcn_ref
from oracle.thirdparty import deform_conv2d


def _case(seed, n, c, cout, G, H, W, dtype, special):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, H, W, generator=g, dtype=dtype)
    w = torch.randn(cout, c, 3, 3, generator=g, dtype=dtype) / (3 * c ** 0.5)
    b = torch.randn(cout, generator=g, dtype=dtype)
    offset = torch.randn(n, 2 * G * 9, H, W, generator=g, dtype=dtype) * 3
    mask = torch.rand(n, G * 9, H, W, generator=g, dtype=dtype)
    if special:
        # sampling position = base (h - 1 + i) + offset: force the positions the reference's rules single out
        off = offset.view(n, G, 9, 2, H, W)
        hh = torch.arange(H, dtype=dtype).view(1, H, 1)
        ww = torch.arange(W, dtype=dtype).view(1, 1, W)
        for k in range(9):
            i, j = divmod(k, 3)
            by, bx = hh - 1 + i, ww - 1 + j
            targets = [(-0.5, 0.25), (-1.0, 2.0), (-0.999, -0.001), (H - 1.0, W - 1.0), (H - 0.5, W - 0.25),
                       (float(H), 1.0), (2.0, 3.0), (H - 1.25, -1.0), (-3.0, 1.5)]
            ty, tx = targets[k]
            # group k % G takes the special target at tap k; other groups stay random
            gsel = k % G
            off[:, gsel, k, 0] = (ty - by).expand(n, H, W)
            off[:, gsel, k, 1] = (tx - bx).expand(n, H, W)
        # exactly-integer offsets on another group
        off[:, (G - 1), :, :] = torch.round(off[:, (G - 1), :, :])
    return x, w, b, offset, mask


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
@pytest.mark.parametrize("special", [False, True])
@pytest.mark.parametrize("shape", [(1, 16, 8, 4, 7, 9), (2, 32, 12, 16, 6, 5)])
def test_thirdparty_deform_conv2d_equals_reference_source(dtype, tol, special, shape):
    n, c, cout, G, H, W = shape
    x, w, b, offset, mask = _case(3 + c, n, c, cout, G, H, W, dtype, special)
    got = deform_conv2d(x, offset, w, b, (1, 1), (1, 1), (1, 1), mask).numpy()
    ref = dcn_ref.modulated_deform_conv_forward(x.numpy(), w.numpy(), b.numpy(), offset.numpy(), mask.numpy(),
                                                deformable_group=G)
    err = np.abs(got - ref).max()
    assert err <= tol * max(1.0, np.abs(ref).max()), f"deform_conv2d vs dcn source: {err:.3e}"


def test_channel_order_of_offsets_is_pinned():
    """Offset channel 2*(g*9+k) is the ROW displacement, +1 the COLUMN displacement, mask channel g*9+k
    (deform_conv_cuda_kernel.cu:600-613): a one-hot input shifted by an integer offset lands where that order says."""
    H = W = 6
    x = np.zeros((1, 2, H, W), dtype=np.float64)
    x[0, 0, 4, 1] = 1.0          # channel 0 (group 0)
    x[0, 1, 2, 5] = 1.0          # channel 1 (group 1)
    w = np.zeros((2, 2, 3, 3), dtype=np.float64)
    w[0, 0, 1, 1] = 1.0          # centre tap (k = 4) only, identity on channels
    w[1, 1, 1, 1] = 1.0
    G = 2
    offset = np.zeros((1, 2 * G * 9, H, W), dtype=np.float64)
    mask = np.ones((1, G * 9, H, W), dtype=np.float64)
    offset[0, 2 * (0 * 9 + 4)] = 2.0       # group 0, centre tap: dy = +2
    offset[0, 2 * (0 * 9 + 4) + 1] = -1.0  #                       dx = -1
    offset[0, 2 * (1 * 9 + 4)] = -1.0      # group 1: dy = -1, dx = +3
    offset[0, 2 * (1 * 9 + 4) + 1] = 3.0
    mask[0, 1 * 9 + 4] = 0.5
    out = dcn_ref.modulated_deform_conv_forward(x, w, None, offset, mask, deformable_group=G)
    # out[p] = x[p + d]: the impulse at (4,1) appears at (4-2, 1+1) = (2,2); the one at (2,5) at (3,2) with mask 0.5
    exp = np.zeros((1, 2, H, W))
    exp[0, 0, 2, 2] = 1.0
    exp[0, 1, 3, 2] = 0.5
    assert np.array_equal(out, exp)
    got = deform_conv2d(torch.from_numpy(x), torch.from_numpy(offset), torch.from_numpy(w), None, (1, 1), (1, 1), (1, 1),
                        torch.from_numpy(mask)).numpy()
    assert np.abs(got - exp).max() < 1e-12
