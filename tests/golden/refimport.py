"""Import the reference's ``guided_diffusion`` package in the build container.

Used ONLY by ``make_golden.py`` (fixture generation; never on the GPU box, where
``/root/reference`` does not exist).  The reference's third-party dependencies
that are absent here are replaced by the restatements in
``oracle/thirdparty.py`` or by empty modules when they are imported but never
called on the sampling path.  The reference's own code runs unmodified.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import scipy.signal
    import scipy.signal.windows
    import torch

    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import thirdparty as tp

    sys.dont_write_bytecode = True

    class FaceRestoreHelper:  # type hint only (gaussian_diffusion.py:12)
        pass

    # the reference package's own facelib pulls cv2 + network weights; only a type
    # hint is needed from it on the sampling path.
    _mod("guided_diffusion.facelib")
    _mod("guided_diffusion.facelib.utils")
    _mod("guided_diffusion.facelib.utils.face_restoration_helper",
         FaceRestoreHelper=FaceRestoreHelper)
    _mod("more_itertools")
    _mod("flash_attn")
    _mod("flash_attn.flash_attn_interface", flash_attn_func=tp.flash_attn_func)
    tv = _mod("torchvision")
    tv.ops = _mod("torchvision.ops", deform_conv2d=tp.deform_conv2d)
    tv.transforms = _mod("torchvision.transforms")
    tv.transforms.functional = _mod(
        "torchvision.transforms.functional",
        normalize=lambda x, mean, std: (x - mean) / std)
    _mod("mmcv")
    _mod("mmcv.cnn", constant_init=tp.constant_init)
    _mod("mmcv.ops", ModulatedDeformConv2d=tp.ModulatedDeformConv2d)
    _mod("mmedit")
    _mod("mmedit.models")
    _mod("mmedit.models.common", flow_warp=tp.flow_warp, PixelShufflePack=tp.PixelShufflePack)
    _mod("mmedit.models.backbones")
    _mod("mmedit.models.backbones.sr_backbones")
    _mod("mmedit.models.backbones.sr_backbones.basicvsr_net",
         ResidualBlocksWithInputConv=tp.ResidualBlocksWithInputConv, SPyNet=tp.SPyNet)
    # cv2: only Cubic_Kernel (imresize_pseudoSR.py:96-102) calls it and its result is
    # discarded for kernel_indx >= 8 (imresize_pseudoSR.py:26-37).
    import numpy as np

    def _resize(img, dsize, interpolation=None):
        out = np.zeros((dsize[1], dsize[0]), dtype=img.dtype)
        c = (np.array(out.shape) - 1) // 2
        out[c[0] - 3:c[0] + 4, c[1] - 3:c[1] + 4] = 1.0  # any non-empty support works
        return out

    _mod("cv2", resize=_resize, INTER_CUBIC=2)
    if not hasattr(scipy.signal, "gaussian"):
        scipy.signal.gaussian = scipy.signal.windows.gaussian
    _mod("superslomo")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def ref(module):
    """Import ``guided_diffusion.<module>`` from the reference tree."""
    return importlib.import_module("guided_diffusion." + module)


class cuda_shaped:
    """Build reference modules with ``torch.cuda.is_available()`` patched True so the
    deformable-alignment branch exists (unet_new.py:650)."""

    def __enter__(self):
        import torch
        self._orig = torch.cuda.is_available
        torch.cuda.is_available = lambda: True

    def __exit__(self, *a):
        import torch
        torch.cuda.is_available = self._orig
