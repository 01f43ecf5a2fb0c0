"""ctypes binding of libflair_hip.so (the C ABI declared in include/flair_hip.h).

The library is the product path: there is no CPU or PyTorch fallback.  Loading fails
loudly (``FlairHipUnavailable``) when the shared object has not been built, and every
wrapper raises ``FlairHipError`` with the library's message on a non-zero status.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: the library binds to torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflair_hip.so")

FLAIR_F32, FLAIR_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU01, ACT_SILU, ACT_DCN_OFFSETS, ACT_LRELU02, ACT_GELU = 0, 1, 2, 3, 4, 5, 6


class FlairHipUnavailable(RuntimeError):
    pass


class FlairHipError(RuntimeError):
    pass


class ConvParams(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int), ("T", ctypes.c_int), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("KT", ctypes.c_int), ("KH", ctypes.c_int),
                ("KW", ctypes.c_int), ("Cout", ctypes.c_int), ("nseg", ctypes.c_int),
                ("seg_c", ctypes.c_int * 4), ("seg_ld", ctypes.c_int * 4),
                ("y_ld", ctypes.c_int), ("res_ld", ctypes.c_int * 2), ("act", ctypes.c_int),
                ("out_scale", ctypes.c_float), ("frame_bias_ld", ctypes.c_int), ("stride", ctypes.c_int),
                ("act_param", ctypes.c_float), ("act_period", ctypes.c_int), ("asym_pad", ctypes.c_int),
                ("reflect_pad", ctypes.c_int)]


_lib = None


def lib():
    """The loaded library; raises FlairHipUnavailable if it has not been built."""
    global _lib
    if _lib is None:
        global LIB_PATH
        if os.environ.get("FLAIR_HIP_LIB"):        # diagnostic builds (make timing / make probe) only
            LIB_PATH = os.environ["FLAIR_HIP_LIB"]
        if not os.path.exists(LIB_PATH):
            raise FlairHipUnavailable(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C flair_amd/csrc`). The HIP library is required; "
                f"there is no fallback path.")
        l = ctypes.CDLL(LIB_PATH)
        l.flair_last_error.restype = ctypes.c_char_p
        _lib = l
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().flair_last_error().decode("utf-8", "replace")
        raise FlairHipError(f"{what} failed (status {status}): {msg}")


def dtype_code(t):
    if t.dtype == torch.float32:
        return FLAIR_F32
    if t.dtype == torch.bfloat16:
        return FLAIR_BF16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise FlairHipError("flair_amd ops need tensors resident in HBM (device='cuda'); "
                            "no CPU path exists")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
