"""flair_amd -- FLAIR's diffusion sampling hot path on MI355X (gfx950).

Layout:
  csrc/               hand-written HIP kernels + the C ABI (include/flair_hip.h)
  _lib.py, ops.py     ctypes binding and tensor-level wrappers
  guided_diffusion/   host-side mirror of the reference's Python interface
  parallel.py         clip-parallel multi-GPU helpers (RCCL weight broadcast)
"""
import os
import sys

__version__ = "0.2.0"


def install_as_guided_diffusion(reference_root=None):
    """Make ``import guided_diffusion.<x>`` bind to the MI355X implementation.

    A package object named ``guided_diffusion`` is registered whose search path is
    ``flair_amd/guided_diffusion`` first and -- when ``reference_root`` (a checkout of
    wustl-cig/FLAIR) is given -- the reference's own ``guided_diffusion`` directory second.
    Every module this package implements (gaussian_diffusion, respace, unet_new, unet, sr3,
    nn, nn_new, script_util, pseudoSR, jpeg, restore_util, resizer, codeformer, ...) is imported
    under its real name and aliased, so its relative imports keep working; everything else the
    script imports (``guided_diffusion.facelib...``, scripts/video_sample.py:28) still resolves to
    the reference's files through the second path entry.  Returns the package object."""
    import importlib
    import pkgutil
    import types

    impl = importlib.import_module("flair_amd.guided_diffusion")
    pkg = types.ModuleType("guided_diffusion", impl.__doc__)
    pkg.__file__ = impl.__file__
    pkg.__package__ = "guided_diffusion"
    pkg.__path__ = list(impl.__path__)
    if reference_root is not None:
        ref = os.path.join(os.fspath(reference_root), "guided_diffusion")
        if not os.path.isdir(ref):
            raise FileNotFoundError(f"install_as_guided_diffusion: {ref} is not a directory")
        pkg.__path__.append(ref)
    # drop stale aliases of an earlier call / of the reference package itself
    for name in [n for n in sys.modules if n == "guided_diffusion" or n.startswith("guided_diffusion.")]:
        del sys.modules[name]
    sys.modules["guided_diffusion"] = pkg
    for info in pkgutil.iter_modules(impl.__path__):
        mod = importlib.import_module("flair_amd.guided_diffusion." + info.name)
        sys.modules["guided_diffusion." + info.name] = mod
        setattr(pkg, info.name, mod)
    return pkg
