"""Helpers shared by the parity tests."""
import torch


def to_clip(x, dtype, device, pad_to=None):
    """(N,C,H,W) f32 cpu -> (N,H,W,C[padded]) clip tensor on device."""
    y = x.permute(0, 2, 3, 1).contiguous()
    if pad_to is not None and pad_to > y.shape[3]:
        y = torch.cat([y, y.new_zeros(*y.shape[:3], pad_to - y.shape[3])], dim=3)
    return y.to(device=device, dtype=dtype).contiguous()


def from_clip(y, c=None):
    y = y.float().cpu()
    if c is not None:
        y = y[..., :c]
    return y.permute(0, 3, 1, 2).contiguous()


def rb(x, dtype):
    """Round an f32 cpu tensor through `dtype` (so the CPU reference sees what the GPU sees)."""
    return x.to(dtype).float()


# Tolerances (stated per dtype, relative to the reference's max magnitude):
#   f32 : accumulation-order differences only            -> 2e-5 * max|ref| + 1e-6
#   bf16: output rounding (2^-9 rel) + bf16 intermediates -> 1.6e-2 * max|ref| + 1e-3
TOL = {torch.float32: (2e-5, 1e-6), torch.bfloat16: (1.6e-2, 1e-3)}


def assert_close(got, ref, dtype, what="", scale=1.0):
    rel, ab = TOL[dtype]
    err = (got - ref).abs().max().item()
    bound = scale * rel * ref.abs().max().item() + ab
    assert err <= bound, f"{what}: max|err|={err:.3e} > {bound:.3e} (max|ref|={ref.abs().max().item():.3e})"
    return err


def parity_log(line):
    """Append one measured-error line to gpurun_out/r04_parity.txt (merged back from the GPU box; the copy under
    profiles/ is the committed record).  Never fails a test."""
    import os
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = os.path.join(root, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "r04_parity.txt"), "a") as f:
            f.write(line.rstrip() + "\n")
    except OSError:
        pass
    print(line)
