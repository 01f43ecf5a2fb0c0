"""Names the reference's ``guided_diffusion/unet.py`` exports to ``sr3.py`` (unet.py:64-77,113-254,
313-595,664-758), bound to the MI355X implementations.  ``CrossFrameUNetModel`` is dead code in
the reference (it calls ``BasicVSRPP`` with a signature that does not exist, unet.py:1152-1165)
and is not provided."""
from .sr3 import FlowVSRPP as BasicVSRPP  # noqa: F401
from .sr3 import ResBlock, TemporalWrapper  # noqa: F401
from .unet_new import SecondOrderDeformableAlignment, TemporalAttention  # noqa: F401
