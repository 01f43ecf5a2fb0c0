"""flair_amd -- FLAIR's diffusion sampling hot path on MI355X (gfx950).

Layout:
  csrc/               hand-written HIP kernels + the C ABI (include/flair_hip.h)
  _lib.py, ops.py     ctypes binding and tensor-level wrappers
  guided_diffusion/   host-side mirror of the reference's Python interface
  parallel.py         clip-parallel multi-GPU helpers (RCCL weight broadcast)
"""
import sys

__version__ = "0.1.0"


def install_as_guided_diffusion():
    """Alias ``flair_amd.guided_diffusion`` as top-level ``guided_diffusion`` so code written
    against the reference (``from guided_diffusion.unet_new import UNetModel`` ...) binds to
    the MI355X implementation."""
    import importlib
    pkg = importlib.import_module("flair_amd.guided_diffusion")
    sys.modules["guided_diffusion"] = pkg
    for name in ("nn_new", "unet_new", "gaussian_diffusion", "respace", "script_util",
                 "pseudoSR", "jpeg", "restore_util", "resizer"):
        try:
            sys.modules["guided_diffusion." + name] = importlib.import_module(
                "flair_amd.guided_diffusion." + name)
        except ModuleNotFoundError:
            pass
    return pkg
