"""Tensor-level wrappers over the C ABI (include/flair_hip.h).

Every function takes torch tensors that live in HBM, passes raw pointers / sizes to
libflair_hip.so on torch's current HIP stream and returns torch tensors.  Activations
are *clip tensors*: shape (T, H, W, C), C innermost, dtype float32 or bfloat16.
PyTorch only provides memory and streams here; no torch operator computes anything.
"""
import ctypes

import torch

from . import _lib
from ._lib import (ACT_DCN_OFFSETS, ACT_GELU, ACT_LRELU01, ACT_LRELU02, ACT_NONE, ACT_RELU, ACT_SILU, ConvParams, check, dtype_code, lib,
                   ptr, stream)

__all__ = ["conv", "conv_chain", "group_norm", "nchw_to_clip", "clip_to_nchw", "timestep_embedding", "linear",
           "qkv_attention", "temporal_attention", "flow_warp", "flow_compose", "resize",
           "dcn_align", "dcn_raw_permutation", "scale_pixels", "predict_xstart", "sampler_update",
           "ACT_NONE", "ACT_RELU", "ACT_LRELU01", "ACT_SILU", "ACT_DCN_OFFSETS", "ACT_LRELU02", "ACT_GELU"]


class GnParams(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int), ("C", ctypes.c_int), ("c0", ctypes.c_int),
                ("ld0", ctypes.c_int), ("ld1", ctypes.c_int), ("groups", ctypes.c_int),
                ("F", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("frames_per_stat", ctypes.c_int), ("eps", ctypes.c_float), ("act", ctypes.c_int),
                ("resample", ctypes.c_int), ("y_ld", ctypes.c_int), ("raw_ld", ctypes.c_int),
                ("film_ld", ctypes.c_int)]


class SamplerCoefs(ctypes.Structure):
    _fields_ = [("gamma", ctypes.c_float), ("w_aux", ctypes.c_float),
                ("sqrt_recip_alphas_cumprod", ctypes.c_float),
                ("sqrt_recipm1_alphas_cumprod", ctypes.c_float),
                ("sqrt_alphas_cumprod_prev", ctypes.c_float),
                ("sqrt_one_minus_alphas_cumprod_prev", ctypes.c_float),
                ("sqrt_one_minus_rho", ctypes.c_float), ("sqrt_rho", ctypes.c_float),
                ("clip_denoised", ctypes.c_int), ("nonzero", ctypes.c_int),
                ("frame_elems", ctypes.c_long), ("frames", ctypes.c_int),
                ("prev_frames", ctypes.c_int)]


class AttnParams(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int), ("frames", ctypes.c_int), ("L", ctypes.c_int),
                ("heads", ctypes.c_int), ("head_dim", ctypes.c_int), ("ld", ctypes.c_int),
                ("out_ld", ctypes.c_int), ("q_off", ctypes.c_int), ("k_off", ctypes.c_int),
                ("v_off", ctypes.c_int), ("head_stride", ctypes.c_int), ("scale", ctypes.c_float)]


class TAttnParams(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int), ("T", ctypes.c_int), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("C", ctypes.c_int), ("window", ctypes.c_int),
                ("ld", ctypes.c_int), ("out_ld", ctypes.c_int), ("round_fp16", ctypes.c_int),
                ("scale", ctypes.c_float)]


class DcnParams(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int), ("F", ctypes.c_int), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("Cin", ctypes.c_int), ("Cout", ctypes.c_int),
                ("G", ctypes.c_int), ("x_ld", ctypes.c_int * 2), ("raw_ld", ctypes.c_int),
                ("y_ld", ctypes.c_int), ("max_residue_magnitude", ctypes.c_float), ("raw_activated", ctypes.c_int)]


def _ld(t):
    """Pixel stride (elements) of a clip tensor or channel-slice view of one."""
    assert t.dim() == 4 and t.stride(3) == 1, "clip tensors are (T,H,W,C) with C innermost"
    ld = t.stride(2)
    assert t.stride(1) == ld * t.shape[2] and (t.shape[0] == 1 or t.stride(0) == ld * t.shape[1] * t.shape[2]), \
        "clip tensor must be dense over (T,H,W)"
    return ld


def _f32(t):
    assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    return t


def k_align(dtype):
    """Channel granularity of a conv input segment (one 64-byte K step)."""
    return 32 if dtype == torch.bfloat16 else 16


def pad_channels(c, dtype):
    g = k_align(dtype)
    return (c + g - 1) // g * g


# --------------------------------------------------------------------------- conv
def conv(xs, weight, bias, cout, kernel, *, out=None, act=ACT_NONE, res0=None, res1=None,
         out_scale=1.0, stride=1, frame_bias=None, act_param=0.0, act_period=0, asym_pad=False, reflect_pad=False):
    """Y = act(conv(cat(xs), W) + bias) + res0 + res1, times out_scale.

    xs: one clip tensor or a list of up to 4 (channel-concatenated implicitly; each
    width a multiple of ``k_align``); weight: packed [cout][taps][sum C] in the
    activation dtype; kernel: (KT, KH, KW); out: optional (T,H,W,>=cout) view.
    """
    if isinstance(xs, torch.Tensor):
        xs = [xs]
    x0 = xs[0]
    T, H, W, _ = x0.shape
    p = ConvParams()
    p.dtype = dtype_code(x0)
    p.T, p.H, p.W = T, H, W
    p.KT, p.KH, p.KW = kernel
    p.Cout = cout
    p.nseg = len(xs)
    arr = (ctypes.c_void_p * 4)()
    for i, x in enumerate(xs):
        assert x.dtype == x0.dtype and x.shape[:3] == x0.shape[:3]
        p.seg_c[i] = x.shape[3]
        p.seg_ld[i] = _ld(x)
        arr[i] = x.data_ptr()
    p.stride = stride
    p.asym_pad = int(asym_pad)
    p.reflect_pad = int(reflect_pad)
    p.frame_bias_ld = frame_bias.stride(0) if frame_bias is not None else 0
    if frame_bias is not None:
        assert frame_bias.dtype == torch.float32 and frame_bias.stride(1) == 1 and frame_bias.shape[0] == T
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    if out is None:
        out = torch.empty((T, Ho, Wo, cout), dtype=x0.dtype, device=x0.device)
    assert tuple(out.shape[:3]) == (T, Ho, Wo) and out.shape[3] >= cout and out.dtype == x0.dtype
    p.y_ld = _ld(out)
    p.res_ld[0] = _ld(res0) if res0 is not None else 0
    p.res_ld[1] = _ld(res1) if res1 is not None else 0
    p.act = act
    p.act_param, p.act_period = act_param, act_period
    p.out_scale = out_scale
    assert weight.dtype == x0.dtype and weight.is_contiguous()
    e0 = _prof_begin()
    wsb = _conv_ws_bytes(p)
    ws = _workspace(wsb, x0.device, "conv") if wsb else None

    def launch(_keep=xs):       # the segment tensors reach the library as raw pointers: keep them alive with the closure
        check(lib().flair_conv_nhwc(ctypes.byref(p), arr, ptr(weight), ptr(_f32(bias)), ptr(frame_bias), ptr(res0),
                                    ptr(res1), ptr(out), ptr(ws), ctypes.c_size_t(wsb), stream()), "flair_conv_nhwc")
    launch()
    if e0 is not None:
        cin = sum(x.shape[3] for x in xs)
        taps = kernel[0] * kernel[1] * kernel[2]
        esz = x0.element_size()
        flops = 2.0 * T * Ho * Wo * cout * cin * taps
        nbytes = esz * (cin * T * H * W + cout * T * Ho * Wo * (1 + (res0 is not None) + (res1 is not None))
                        + cout * taps * cin)
        _prof_end(e0, ("conv", lib().flair_conv_variant(ctypes.byref(p))), x0.dtype, flops, nbytes, launch,
                  ("conv", T, H, W, tuple(x.shape[3] for x in xs), cout, tuple(kernel), stride, act,
                   res0 is not None, res1 is not None, frame_bias is not None))
    return out


# bench.py sets PROFILE to a list to bracket every conv / GroupNorm / alignment / attention call with HIP
# events on the launch stream (roofline leg): entries are (family, dtype name, algorithmic FLOPs,
# algorithmic bytes, start event, end event).
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_end(e0, family, dtype, flops, nbytes, replay=None, sig=None):
    """replay: closure that re-issues the identical library call (its arguments stay alive with it), so that bench.py can
    time every distinct launch shape by hipGraph replay of back-to-back launches; sig: hashable shape signature."""
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.append((family, str(dtype), float(flops), float(nbytes), e0, e1, replay, sig))


def conv_variant(T, H, W, cins, cout, kernel, dtype=torch.bfloat16, stride=1):
    """Kernel variant flair_conv_nhwc dispatches this geometry to (flair_conv_variant); host-only."""
    p = ConvParams()
    p.dtype = 1 if dtype == torch.bfloat16 else 0      # FLAIR_BF16 / FLAIR_F32
    p.T, p.H, p.W = T, H, W
    p.KT, p.KH, p.KW = kernel
    p.Cout = cout
    p.nseg = len(cins)
    for i, c in enumerate(cins):
        p.seg_c[i] = c
        p.seg_ld[i] = c
    p.stride = stride
    p.y_ld = cout
    return lib().flair_conv_variant(ctypes.byref(p))


def _conv_ws_bytes(p):
    f = lib().flair_conv_workspace_bytes
    f.restype = ctypes.c_size_t
    return f(ctypes.byref(p))


def pack_conv_weight(w, seg_channels, dtype, cout_pad=None):
    """Reference layout (Cout, Cin, *k) f32 -> [Cout_pad][taps][sum padded seg] in `dtype`.

    ``seg_channels``: list of (real, padded) channel counts of the input segments, in
    concatenation order.  One-time repack at load (plain tensor plumbing).
    """
    cout, cin = w.shape[:2]
    taps = 1
    for k in w.shape[2:]:
        taps *= k
    w = w.reshape(cout, cin, taps).permute(0, 2, 1).float()  # cout, taps, cin
    parts, o = [], 0
    for real, padded in seg_channels:
        seg = w[:, :, o:o + real]
        if padded > real:
            seg = torch.cat([seg, seg.new_zeros(cout, taps, padded - real)], dim=2)
        parts.append(seg)
        o += real
    assert o == cin, (o, cin)
    w = torch.cat(parts, dim=2)
    if cout_pad is not None and cout_pad > cout:
        w = torch.cat([w, w.new_zeros(cout_pad - cout, taps, w.shape[2])], dim=0)
    return w.to(dtype).contiguous()


# ---------------------------------------------------------------- fused conv chains
class ChainParams(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int), ("T", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("C", ctypes.c_int), ("CoutB", ctypes.c_int), ("nseg", ctypes.c_int),
                ("seg_c", ctypes.c_int * 4), ("seg_ld", ctypes.c_int * 4), ("y_ld", ctypes.c_int),
                ("res_ld", ctypes.c_int * 2), ("actA", ctypes.c_int), ("actB", ctypes.c_int),
                ("out_scale", ctypes.c_float), ("act_param", ctypes.c_float), ("act_period", ctypes.c_int)]


def chain_supported(xs, c_mid):
    """Where the fused chain kernel is used: the geometry flair_conv_chain covers (W a multiple of 8, 64 or 128
    mid channels) AND a launch of at most ~2 workgroups per CU (per-frame calls of the recurrence).  Clip-level
    calls (thousands of tiles) stay on the two-launch path: there independent workgroups, two per CU, already
    overlap each other's fetch / MFMA / store phases, which one 135 KB workgroup per CU cannot."""
    x0 = xs[0] if isinstance(xs, (list, tuple)) else xs
    T, H, W, _ = x0.shape
    tiles = T * ((H + 7) // 8) * (W // 32 if W % 32 == 0 and T * ((H + 7) // 8) * (W // 32) >= 192 else W // 8)
    return c_mid in (64, 128) and W % 8 == 0 and tiles <= 512


def conv_chain(xs, wA, bA, actA, wB, bB, actB, c_mid, coutB, *, res0=None, res1=None, out_scale=1.0, out=None,
               act_param=0.0, act_period=0):
    """Y = (actB(conv3x3(actA(conv3x3(cat(xs), wA) + bA), wB) + bB) + res0 + res1) * out_scale in ONE launch
    (the c_mid-channel intermediate stays in LDS).  wA=None: no first stage; xs is the c_mid-channel input,
    staged once and kept resident while coutB output channels are produced (wide-output convolutions)."""
    if isinstance(xs, torch.Tensor):
        xs = [xs]
    x0 = xs[0]
    T, H, W, _ = x0.shape
    p = ChainParams()
    p.dtype = dtype_code(x0)
    p.T, p.H, p.W = T, H, W
    p.C, p.CoutB = c_mid, coutB
    p.nseg = len(xs)
    arr = (ctypes.c_void_p * 4)()
    for i, x in enumerate(xs):
        assert x.dtype == x0.dtype and x.shape[:3] == x0.shape[:3]
        p.seg_c[i] = x.shape[3]
        p.seg_ld[i] = _ld(x)
        arr[i] = x.data_ptr()
    if out is None:
        out = torch.empty((T, H, W, coutB), dtype=x0.dtype, device=x0.device)
    assert tuple(out.shape[:3]) == (T, H, W) and out.shape[3] >= coutB and out.dtype == x0.dtype
    p.y_ld = _ld(out)
    p.res_ld[0] = _ld(res0) if res0 is not None else 0
    p.res_ld[1] = _ld(res1) if res1 is not None else 0
    p.actA, p.actB = actA, actB
    p.act_param, p.act_period = act_param, act_period
    p.out_scale = out_scale
    assert wB.dtype == x0.dtype and wB.is_contiguous() and (wA is None or (wA.dtype == x0.dtype and wA.is_contiguous()))
    e0 = _prof_begin()

    def launch(_keep=xs):
        check(lib().flair_conv_chain(ctypes.byref(p), arr, ptr(wA), ptr(_f32(bA)), ptr(wB), ptr(_f32(bB)), ptr(res0),
                                     ptr(res1), ptr(out), stream()), "flair_conv_chain")
    launch()
    if e0 is not None:
        cin = sum(x.shape[3] for x in xs)
        esz = x0.element_size()
        px = T * H * W
        flops = 2.0 * 9 * px * (c_mid * coutB + (cin * c_mid if wA is not None else 0))
        nbytes = esz * (px * (cin + coutB * (1 + (res0 is not None) + (res1 is not None)))
                        + 9 * c_mid * coutB + (9 * cin * c_mid if wA is not None else 0))
        _prof_end(e0, ("chain", 2 if wA is not None else 1, c_mid), x0.dtype, flops, nbytes, launch,
                  ("chain", T, H, W, tuple(x.shape[3] for x in xs), c_mid, coutB, wA is not None, res0 is not None,
                   res1 is not None))
    return out


# --------------------------------------------------------------------- group norm
_gn_ws = {}


def _workspace(nbytes, device, tag="gn"):
    key = (tag, device, torch.cuda.current_stream().cuda_stream)
    buf = _gn_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _gn_ws[key] = buf
    return buf


def group_norm(x, gamma, beta, *, x1=None, groups=32, eps=1e-5, act=ACT_NONE, film=None,
               frames_per_stat=None, resample=0, out=None, want_raw=False):
    """act(GN(cat(x, x1)) * (1+scale) + shift) with optional 2x resampling.

    film: (F, >=2C) f32 rows (scale | shift) or None.  resample: 0 / 1 (avg-pool 2) /
    2 (nearest x2).  Returns y, or (y, raw) when ``want_raw``.
    """
    T, H, W, c0 = x.shape
    C = c0 + (x1.shape[3] if x1 is not None else 0)
    p = GnParams()
    p.dtype = dtype_code(x)
    p.C, p.c0 = C, c0
    p.ld0 = _ld(x)
    p.ld1 = _ld(x1) if x1 is not None else 0
    p.groups = groups
    p.F, p.H, p.W = T, H, W
    p.frames_per_stat = frames_per_stat or T
    p.eps = eps
    p.act = act
    p.resample = resample
    Ho, Wo = (H // 2, W // 2) if resample == 1 else ((H * 2, W * 2) if resample == 2 else (H, W))
    if out is None:
        out = torch.empty((T, Ho, Wo, C), dtype=x.dtype, device=x.device)
    raw = torch.empty((T, Ho, Wo, C), dtype=x.dtype, device=x.device) if want_raw else None
    p.y_ld = _ld(out)
    p.raw_ld = _ld(raw) if raw is not None else 0
    p.film_ld = film.stride(0) if film is not None else 0
    if film is not None:
        assert film.dtype == torch.float32 and film.stride(1) == 1 and film.shape[0] == T
    ws = _workspace(lib_ws_bytes(p), x.device)
    e0 = _prof_begin()

    def launch():
        check(lib().flair_groupnorm_nhwc(ctypes.byref(p), ptr(x), ptr(x1), ptr(_f32(gamma)), ptr(_f32(beta)),
                                         ptr(film), ptr(out), ptr(raw), ptr(ws), stream()),
              "flair_groupnorm_nhwc")
    launch()
    if e0 is not None:   # two-pass GroupNorm: the input is read twice, the output written once
        numel_in, numel_out = T * H * W * C, T * Ho * Wo * C
        _prof_end(e0, ("gn",), x.dtype, 8.0 * numel_in,
                  x.element_size() * (2 * numel_in + numel_out * (2 if want_raw else 1)), launch,
                  ("gn", T, H, W, C, c0, groups, p.frames_per_stat, act, resample, want_raw, film is not None))
    return (out, raw) if want_raw else out


def lib_ws_bytes(p):
    f = lib().flair_groupnorm_workspace_bytes
    f.restype = ctypes.c_size_t
    return f(ctypes.byref(p))


# ------------------------------------------------------------------ edge / small ops
def nchw_to_clip(src, dst, coff=0):
    """(N,C,H,W) f32 -> channels [coff, coff+C) of clip tensor dst (N,H,W,Cd)."""
    N, C, H, W = src.shape
    assert src.dtype == torch.float32 and src.is_contiguous()
    check(lib().flair_nchw_f32_to_nhwc(ptr(src), N, C, H, W, ptr(dst), dtype_code(dst), _ld(dst), coff,
                                       stream()), "flair_nchw_f32_to_nhwc")
    return dst


def clip_to_nchw(src, C, coff=0, out=None):
    N, H, W, _ = src.shape
    if out is None:
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=src.device)
    check(lib().flair_nhwc_to_nchw_f32(ptr(src), dtype_code(src), _ld(src), coff, N, C, H, W, ptr(out),
                                       stream()), "flair_nhwc_to_nchw_f32")
    return out


def timestep_embedding(t, dim, max_period=10000.0, out=None, sin_first=False):
    """t: (N,) f32 device tensor -> (N, dim) f32 ([cos|sin], or [sin|cos] when sin_first)."""
    assert t.dtype == torch.float32 and t.is_contiguous()
    N = t.shape[0]
    if out is None:
        out = torch.empty((N, dim), dtype=torch.float32, device=t.device)
    check(lib().flair_timestep_embedding(ptr(t), N, dim, ctypes.c_float(max_period), int(sin_first), ptr(out),
                                         stream()), "flair_timestep_embedding")
    return out


def linear(x, w, b, *, act_in=ACT_NONE, act_out=ACT_NONE, out=None):
    """f32 (M<=32, K) @ (N, K)^T + b."""
    M, K = x.shape
    N = w.shape[0]
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(lib().flair_linear_f32(ptr(x), M, K, ptr(w), ptr(b), N, act_in, act_out, ptr(out), out.stride(0),
                                 stream()), "flair_linear_f32")
    return out


def qkv_attention(qkv, heads, *, new_order=False, out=None):
    """qkv: (F, H, W, 3C) clip tensor -> (F, H, W, C)."""
    F_, H, W, C3 = qkv.shape
    C = C3 // 3
    p = AttnParams()
    p.dtype = dtype_code(qkv)
    p.frames, p.L, p.heads, p.head_dim = F_, H * W, heads, C // heads
    p.ld = _ld(qkv)
    if out is None:
        out = torch.empty((F_, H, W, C), dtype=qkv.dtype, device=qkv.device)
    p.out_ld = _ld(out)
    d = C // heads
    if new_order:
        p.q_off, p.k_off, p.v_off, p.head_stride = 0, C, 2 * C, d
    else:
        p.q_off, p.k_off, p.v_off, p.head_stride = 0, d, 2 * d, 3 * d
    p.scale = 1.0 / (d ** 0.5)
    e0 = _prof_begin()

    def launch():
        check(lib().flair_qkv_attention(ctypes.byref(p), ptr(qkv), ptr(out), stream()), "flair_qkv_attention")
    launch()
    if e0 is not None:   # QK^T + AV: 4 * frames * heads * L^2 * d FLOPs; q, k, v read + out written
        L = H * W
        _prof_end(e0, ("attn", L), qkv.dtype, 4.0 * F_ * heads * L * L * d, qkv.element_size() * 4.0 * F_ * L * C,
                  launch, ("attn", F_, L, heads, d, new_order))
    return out


def temporal_attention(qkv, kpos, window, *, round_fp16, out=None):
    T, H, W, C3 = qkv.shape
    C = C3 // 3
    p = TAttnParams()
    p.dtype = dtype_code(qkv)
    p.T, p.H, p.W, p.C, p.window = T, H, W, C, window
    p.ld = _ld(qkv)
    if out is None:
        out = torch.empty((T, H, W, C), dtype=qkv.dtype, device=qkv.device)
    p.out_ld = _ld(out)
    p.round_fp16 = int(round_fp16)
    p.scale = 1.0 / (64 ** 0.5)
    assert kpos.shape == (window - 1, C)
    check(lib().flair_temporal_attention(ctypes.byref(p), ptr(qkv), ptr(_f32(kpos)), ptr(out), stream()),
          "flair_temporal_attention")
    return out


def flow_warp(x, flow, *, border=False, out=None):
    """x: (F,H,W,C) clip tensor; flow: (F,H,W,2) f32 (dx,dy)."""
    F_, H, W, C = x.shape
    assert flow.dtype == torch.float32 and flow.shape == (F_, H, W, 2)
    if out is None:
        out = torch.empty((F_, H, W, C), dtype=x.dtype, device=x.device)
    check(lib().flair_flow_warp(ptr(x), dtype_code(x), _ld(x), ptr(flow), _ld(flow), F_, H, W, C, int(border),
                                ptr(out), _ld(out), stream()), "flair_flow_warp")
    return out


def flow_compose(f1, f2, out=None):
    F_, H, W, _ = f1.shape
    assert f1.dtype == torch.float32 and f1.is_contiguous() and f2.is_contiguous()
    if out is None:
        out = torch.empty_like(f1)
    check(lib().flair_flow_compose(ptr(f1), ptr(f2), F_, H, W, ptr(out), stream()), "flair_flow_compose")
    return out


RESIZE_BILINEAR, RESIZE_BILINEAR_AC, RESIZE_BICUBIC, RESIZE_AVGPOOL2, RESIZE_NEAREST = 0, 1, 2, 3, 4


def resize(x, size, mode, *, channels=None, out=None, scale_c0=1.0, scale_c1=1.0):
    F_, Hi, Wi, Cx = x.shape
    C = channels or Cx
    Ho, Wo = size
    if out is None:
        out = torch.zeros((F_, Ho, Wo, Cx), dtype=x.dtype, device=x.device)
    check(lib().flair_resize_nhwc(ptr(x), dtype_code(x), _ld(x), F_, Hi, Wi, C, mode, Ho, Wo, ptr(out), _ld(out),
                                  ctypes.c_float(scale_c0), ctypes.c_float(scale_c1), stream()),
          "flair_resize_nhwc")
    return out


def dcn_raw_permutation(groups):
    """Index tensor taking the reference's conv_offset channel order (o1 | o2 | mask with
    group-major (g*9+k) indexing, unet_new.py:877-885) to the tap-major order flair_dcn_align
    reads (tap k owns channels [3Gk, 3G(k+1))): new[3Gk + 2g + e] = old[2*(g*9+k)+e],
    new[3Gk + 2G + g] = old[18G + g*9 + k].
    Applied to the OUTPUT channels of the last conv_offset convolution when it is packed."""
    G = groups
    perm = torch.empty(27 * G, dtype=torch.long)
    for k in range(9):
        for g in range(G):
            for e in range(2):
                perm[3 * G * k + 2 * g + e] = 2 * (g * 9 + k) + e
            perm[3 * G * k + 2 * G + g] = 18 * G + g * 9 + k
    return perm


def dcn_align(x0, x1, raw, flow1, flow2, weight, bias, cout, *, groups=16, max_mag=10.0, out=None, raw_activated=False):
    """raw: conv_offset output in TAP-MAJOR channel order (see dcn_raw_permutation); raw_activated: the producing
    convolution already applied ACT_DCN_OFFSETS (max_mag * tanh on the residues, sigmoid on the masks)."""
    F_, H, W, ch = x0.shape
    p = DcnParams()
    p.dtype = dtype_code(x0)
    p.F, p.H, p.W = F_, H, W
    p.Cin, p.Cout, p.G = 2 * ch, cout, groups
    p.x_ld[0], p.x_ld[1] = _ld(x0), _ld(x1)
    p.raw_ld = _ld(raw)
    if out is None:
        out = torch.empty((F_, H, W, cout), dtype=x0.dtype, device=x0.device)
    p.y_ld = _ld(out)
    p.max_residue_magnitude = max_mag
    p.raw_activated = int(raw_activated)
    e0 = _prof_begin()

    def launch():
        check(lib().flair_dcn_align(ctypes.byref(p), ptr(x0), ptr(x1), ptr(raw), ptr(flow1), ptr(flow2), ptr(weight),
                                    ptr(_f32(bias)), ptr(out), stream()), "flair_dcn_align")
    launch()
    if e0 is not None:   # bytes = esz*(2c + 27G + c)*H*W; FLOPs = 2*9*2c*c*H*W + 9*2c*H*W*8
        px = F_ * H * W
        _prof_end(e0, ("dcn", cout), x0.dtype, px * (2.0 * 9 * 2 * ch * cout + 9.0 * 2 * ch * 8),
                  x0.element_size() * px * (2 * ch + 27 * groups + cout), launch,
                  ("dcn", F_, H, W, ch, cout, groups, flow2 is not None))
    return out


def scale_pixels(x, wmap):
    T, H, W, C = x.shape
    check(lib().flair_scale_pixels(ptr(x), dtype_code(x), _ld(x), C, ctypes.c_long(T * H * W), ptr(_f32(wmap)),
                                   stream()), "flair_scale_pixels")
    return x


# ------------------------------------------------------------------------ sampler
def predict_xstart(x, model_out, c_recip, c_recipm1, clip, out=None):
    N, C, H, W = x.shape
    Cm = model_out.shape[1]
    assert x.dtype == torch.float32 and x.is_contiguous() and model_out.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(lib().flair_predict_xstart(ptr(x), ptr(model_out), N, C, Cm, H, W, ctypes.c_float(c_recip),
                                     ctypes.c_float(c_recipm1), int(clip), ptr(out), stream()),
          "flair_predict_xstart")
    return out


def sampler_update(coefs, x, x0, restored, aux, z, prev_recon, out=None):
    if out is None:
        out = torch.empty_like(x)
    for t in (x, x0, restored, aux, z, prev_recon):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    check(lib().flair_sampler_update(ctypes.byref(coefs), ptr(x), ptr(x0), ptr(restored), ptr(aux), ptr(z),
                                     ptr(prev_recon), ctypes.c_long(x.numel()), ptr(out), stream()),
          "flair_sampler_update")
    return out


def affine_channels(x, C, a, b, lo, hi, sub, mul, out):
    """f32 clip tensors: out = (clamp(x*a+b, lo, hi) - sub[c]) * mul[c] on the first C channels."""
    T, H, W, _ = x.shape
    assert x.dtype == torch.float32 and out.dtype == torch.float32
    check(lib().flair_affine_channels_f32(ptr(x), _ld(x), C, ctypes.c_long(T * H * W), ctypes.c_float(a),
                                          ctypes.c_float(b), ctypes.c_float(lo), ctypes.c_float(hi), ptr(sub),
                                          ptr(mul), ptr(out), _ld(out), stream()), "flair_affine_channels_f32")
    return out


def add_frame_bias(x, bias):
    T, H, W, C = x.shape
    assert bias.dtype == torch.float32 and bias.stride(1) == 1
    check(lib().flair_add_frame_bias(ptr(x), dtype_code(x), _ld(x), C, T, ctypes.c_long(H * W), ptr(bias),
                                     bias.stride(0), stream()), "flair_add_frame_bias")
    return x


def cast_channels(src, dst, coff=0):
    """src: (T,H,W,C) f32 -> channels [coff, coff+C) of clip tensor dst."""
    T, H, W, C = src.shape
    assert src.dtype == torch.float32
    check(lib().flair_cast_channels(ptr(src), _ld(src), C, ctypes.c_long(T * H * W), ptr(dst), dtype_code(dst),
                                    _ld(dst), coff, stream()), "flair_cast_channels")
    return dst


def axpby(x, y, a, b, out=None, lo=float("-inf"), hi=float("inf")):
    """out = clamp(a*x + b*y, lo, hi) on flat f32 tensors (y may be None)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and (y is None or y.is_contiguous())
    if out is None:
        out = torch.empty_like(x)
    check(lib().flair_axpby_f32(ptr(x), ptr(y), ctypes.c_float(a), ctypes.c_float(b), ctypes.c_float(lo),
                                ctypes.c_float(hi), ctypes.c_long(x.numel()), ptr(out), stream()),
          "flair_axpby_f32")
    return out


def learned_range_variance(model_out, C, min_log, max_log):
    N, C2, H, W = model_out.shape
    var = torch.empty((N, C, H, W), dtype=torch.float32, device=model_out.device)
    logvar = torch.empty_like(var)
    check(lib().flair_learned_range_variance(ptr(model_out), N, C, H, W, ctypes.c_float(min_log),
                                             ctypes.c_float(max_log), ptr(var), ptr(logvar), stream()),
          "flair_learned_range_variance")
    return var, logvar


# ------------------------------------------------------------------- degradation ops
def depthwise_filter(x, filt, *, pad, out_stride=1, out_offset=0, stuff=1, stuff_offset=0, out_hw=None,
                     reflect=False):
    """x: (N,C,H,W) f32; filt: (kh,kw) f32 device tensor shared by all planes."""
    N, C, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and filt.dtype == torch.float32 and filt.is_contiguous()
    Ho, Wo = out_hw
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    check(lib().flair_depthwise_filter(ptr(x), N * C, H, W, ptr(filt), filt.shape[0], filt.shape[1], pad,
                                       out_stride, out_offset, stuff, stuff_offset, Ho, Wo, int(reflect), ptr(out),
                                       stream()),
          "flair_depthwise_filter")
    return out


def jpeg_roundtrip(x, q_luma, q_chroma, dct8, want_levels=False):
    """x: (N,3,S,S) f32 in [-1,1]; tables: python sequences / numpy arrays of 64 floats (host).
    want_levels: also return the quantised integer levels [luma (N,1,S,S), chroma (N,2,S/2,S/2)] (f32)."""
    N, C, S, _ = x.shape
    assert C == 3 and x.dtype == torch.float32 and x.is_contiguous()
    arr = ctypes.c_float * 64
    ws = torch.empty_like(x)
    out = torch.empty_like(x)
    luma = torch.empty((N, 1, S, S), dtype=torch.float32, device=x.device) if want_levels else None
    chroma = torch.empty((N, 2, S // 2, S // 2), dtype=torch.float32, device=x.device) if want_levels else None
    check(lib().flair_jpeg_roundtrip(ptr(x), N, S, arr(*[float(v) for v in q_luma]),
                                     arr(*[float(v) for v in q_chroma]), arr(*[float(v) for v in dct8]),
                                     ptr(ws), ptr(out), ptr(luma), ptr(chroma), stream()), "flair_jpeg_roundtrip")
    return (out, [luma, chroma]) if want_levels else out


def matmul(a, b):
    """Batched f32 matmul; a: (B,M,K) or (M,K) shared; b: (B,K,N) or (K,N) shared."""
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
    batch = a.shape[0] if a.dim() == 3 else (b.shape[0] if b.dim() == 3 else 1)
    M, K = a.shape[-2:]
    N = b.shape[-1]
    assert b.shape[-2] == K
    out = torch.empty((batch, M, N), dtype=torch.float32, device=a.device)
    check(lib().flair_matmul_f32(ptr(a), ctypes.c_long(M * K if a.dim() == 3 else 0), ptr(b),
                                 ctypes.c_long(K * N if b.dim() == 3 else 0), ptr(out), batch, M, N, K, stream()),
          "flair_matmul_f32")
    return out


def gather_mac(x, outer, lin, inner, fov, w):
    """Resizer step: x viewed [outer][lin][inner] f32; fov int32 (taps,Lout); w f32 (taps,Lout)."""
    taps, lout = fov.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and fov.dtype == torch.int32 and w.dtype == torch.float32
    out = torch.empty((outer, lout, inner), dtype=torch.float32, device=x.device)
    check(lib().flair_gather_mac_f32(ptr(x), ctypes.c_long(outer), lin, ctypes.c_long(inner), ptr(fov.contiguous()),
                                     ptr(w.contiguous()), taps, lout, ptr(out), stream()), "flair_gather_mac_f32")
    return out


def vsrpp_prep(prop, feat2, flow1, flow_prev, cond1, cond2, flow2_out, flowpad):
    """Fused alignment inputs of one BasicVSR++ step (see flair_vsrpp_prep).  One frame:
    prop/feat2/cond*: (1,H,W,c) clip tensors; flows: (1,H,W,2) f32; flowpad: (1,H,W,>=4)."""
    _, H, W, C = prop.shape
    second = flow_prev is not None
    e0 = _prof_begin()

    def launch():
        check(lib().flair_vsrpp_prep(ptr(prop), _ld(prop), ptr(feat2 if second else None),
                                     _ld(feat2) if second else 0, ptr(_f32(flow1)), ptr(_f32(flow_prev)),
                                     dtype_code(prop), H, W, C, ptr(cond1), _ld(cond1), ptr(cond2 if second else None),
                                     _ld(cond2) if second else 0, ptr(flow2_out if second else None), ptr(flowpad),
                                     _ld(flowpad), stream()), "flair_vsrpp_prep")
    launch()
    if e0 is not None:   # reads 1-2 features, writes 1-2 warped features + the 4-channel flow pad
        n = 2 if second else 1
        _prof_end(e0, ("prep",), prop.dtype, 0.0, prop.element_size() * H * W * (2.0 * n * C + flowpad.shape[3]) + 8.0 * H * W * n,
                  launch, ("prep", H, W, C, second))


def vsrpp_warp2(prop, feat2, flow1, flow2, cond1, cond2):
    """cond1 = warp(prop, flow1), cond2 = warp(feat2, flow2) in one launch (flair_vsrpp_warp2); flows precomposed."""
    _, H, W, C = prop.shape
    second = flow2 is not None
    e0 = _prof_begin()

    def launch():
        check(lib().flair_vsrpp_warp2(ptr(prop), _ld(prop), ptr(feat2 if second else None), _ld(feat2) if second else 0,
                                      ptr(_f32(flow1)), ptr(_f32(flow2) if second else None), dtype_code(prop), H, W, C,
                                      ptr(cond1), _ld(cond1), ptr(cond2 if second else None), _ld(cond2) if second else 0,
                                      stream()), "flair_vsrpp_warp2")
    launch()
    if e0 is not None:
        n = 2 if second else 1
        _prof_end(e0, ("prep",), prop.dtype, 0.0, prop.element_size() * H * W * 2.0 * n * C + 8.0 * H * W * n,
                  launch, ("warp2", H, W, C, second))


def add_act(x0, x1=None, act=ACT_NONE, out=None):
    """act(x0 + x1) on clip tensors (x1 may be None)."""
    T, H, W, C = x0.shape
    if out is None:
        out = torch.empty((T, H, W, C), dtype=x0.dtype, device=x0.device)
    check(lib().flair_add_act_nhwc(ptr(x0), _ld(x0), ptr(x1), _ld(x1) if x1 is not None else 0, dtype_code(x0), C,
                                   ctypes.c_long(T * H * W), act, ptr(out), _ld(out), stream()), "flair_add_act_nhwc")
    return out


def maxpool3x3s2(x, out=None):
    """nn.MaxPool2d(3, 2, 1) on a clip tensor."""
    T, H, W, C = x.shape
    if out is None:
        out = torch.empty((T, (H + 1) // 2, (W + 1) // 2, C), dtype=x.dtype, device=x.device)
    check(lib().flair_maxpool3x3s2_nhwc(ptr(x), _ld(x), dtype_code(x), T, H, W, C, ptr(out), _ld(out), stream()),
          "flair_maxpool3x3s2_nhwc")
    return out


# --------------------------------------------------------------------------- calibration launches (measurement only)
def probe_matrix_rate(iters, workgroups_per_cu=1, sink=None):
    """One launch of independent bf16 MFMA chains out of registers on every CU (flair_probe_matrix_rate); returns its FLOPs.
    The caller times it with events on the current stream."""
    if sink is None:
        sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    flop = ctypes.c_double(0.0)
    check(lib().flair_probe_matrix_rate(int(iters), int(workgroups_per_cu), ptr(sink), ctypes.byref(flop), stream()),
          "flair_probe_matrix_rate")
    return flop.value


def probe_stream_rate(src, dst, mode):
    """mode 0: read src, 1: write dst, 2: copy src -> dst (uint8 / any contiguous device tensors of equal size); returns the bytes moved."""
    n = dst.numel() * dst.element_size()
    if mode != 1:
        assert src.numel() * src.element_size() == n
    check(lib().flair_probe_stream_rate(ptr(src) if mode != 1 else ctypes.c_void_p(0), ptr(dst), ctypes.c_size_t(n), int(mode), stream()),
          "flair_probe_stream_rate")
    return n * (2 if mode == 2 else 1)


def gated_blend(x, m, gate, out=None):
    """x + sigmoid(gate[f, c]) * (m - x); gate: (F, >=C) f32 logits."""
    T, H, W, C = x.shape
    assert gate.dtype == torch.float32 and gate.stride(1) == 1 and gate.shape[0] == T
    if out is None:
        out = torch.empty_like(x)
    check(lib().flair_gated_blend(ptr(x), _ld(x), ptr(m), _ld(m), ptr(gate), gate.stride(0), dtype_code(x), C, T,
                                  ctypes.c_long(H * W), ptr(out), _ld(out), stream()), "flair_gated_blend")
    return out


# --------------------------------------------------------------------------- CodeFormer prior pieces
def layer_norm(x, gamma, beta, *, eps=1e-5, pos=None, out=None, out_pos=None):
    """nn.LayerNorm over the channels of every pixel of a clip tensor.  With ``pos`` ((rows_per_frame, C) f32)
    also returns y + pos (broadcast over frames): -> y or (y, y_pos)."""
    F_, H, W, C = x.shape
    if out is None:
        out = torch.empty((F_, H, W, C), dtype=x.dtype, device=x.device)
    if pos is not None and out_pos is None:
        out_pos = torch.empty((F_, H, W, C), dtype=x.dtype, device=x.device)
    if pos is not None:
        assert pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape == (H * W, C)
    check(lib().flair_layernorm_nhwc(ptr(x), dtype_code(x), _ld(x), ctypes.c_long(F_ * H * W), C, ptr(_f32(gamma)),
                                     ptr(_f32(beta)), ctypes.c_float(eps), ptr(out), _ld(out), ptr(pos),
                                     H * W if pos is not None else 0, ptr(out_pos), _ld(out_pos) if pos is not None else 0,
                                     stream()), "flair_layernorm_nhwc")
    return out if pos is None else (out, out_pos)


def attention_wide(qkv, heads, head_dim, *, q_off, k_off, v_off, head_stride, out=None):
    """Attention over the H*W pixels of each frame with heads of any width (flair_attention_wide)."""
    F_, H, W, _ = qkv.shape
    p = AttnParams()
    p.dtype = dtype_code(qkv)
    p.frames, p.L, p.heads, p.head_dim = F_, H * W, heads, head_dim
    p.ld = _ld(qkv)
    if out is None:
        out = torch.empty((F_, H, W, heads * head_dim), dtype=qkv.dtype, device=qkv.device)
    p.out_ld = _ld(out)
    p.q_off, p.k_off, p.v_off, p.head_stride = q_off, k_off, v_off, head_stride
    p.scale = 1.0 / (head_dim ** 0.5)
    check(lib().flair_attention_wide(ctypes.byref(p), ptr(qkv), ptr(out), stream()), "flair_attention_wide")
    return out


def argmax_codebook(logits, n_codes, codebook, *, forced_idx=None, out=None):
    """logits: (F,H,W,>=n_codes) clip tensor; codebook (n_codes, D) f32 -> (codes (F,H,W,D), idx (F*H*W,) int32)."""
    F_, H, W, _ = logits.shape
    D = codebook.shape[1]
    assert codebook.dtype == torch.float32 and codebook.is_contiguous() and codebook.shape[0] == n_codes
    if out is None:
        out = torch.empty((F_, H, W, D), dtype=logits.dtype, device=logits.device)
    idx = torch.empty((F_ * H * W,), dtype=torch.int32, device=logits.device)
    if forced_idx is not None:
        assert forced_idx.dtype == torch.int32 and forced_idx.is_contiguous() and forced_idx.numel() == idx.numel()
    check(lib().flair_argmax_codebook(ptr(logits), dtype_code(logits), _ld(logits), ctypes.c_long(F_ * H * W), n_codes,
                                      ptr(codebook), D, ptr(forced_idx), ptr(idx), ptr(out), _ld(out), stream()),
          "flair_argmax_codebook")
    return out, idx


def adain(content, style, *, eps=1e-5, out=None):
    F_, H, W, C = content.shape
    assert style.shape == content.shape and style.dtype == content.dtype
    if out is None:
        out = torch.empty((F_, H, W, C), dtype=content.dtype, device=content.device)
    check(lib().flair_adain_nhwc(ptr(content), _ld(content), ptr(style), _ld(style), dtype_code(content), F_, H * W, C,
                                 ctypes.c_float(eps), ptr(out), _ld(out), stream()), "flair_adain_nhwc")
    return out


def sft_fuse(dec, scale, shift, w, out=None):
    """dec + w * (dec * scale + shift) on dense clip tensors."""
    assert dec.is_contiguous() and scale.is_contiguous() and shift.is_contiguous()
    assert dec.shape == scale.shape == shift.shape and dec.dtype == scale.dtype == shift.dtype
    if out is None:
        out = torch.empty_like(dec)
    check(lib().flair_sft_fuse(ptr(dec), ptr(scale), ptr(shift), ctypes.c_float(w), dtype_code(dec),
                               ctypes.c_long(dec.numel()), ptr(out), stream()), "flair_sft_fuse")
    return out


# --------------------------------------------------------------------------- un-aligned prior branch (face crop / paste)
def warp_affine_cubic(src, minv, out_hw, *, border=(0.0, 0.0, 0.0), pre=False, post=False):
    """cv2.warpAffine(INTER_CUBIC, BORDER_CONSTANT) per image (flair_warp_affine_cubic).  src: (N,C,Hs,Ws) f32 or f64
    (masks, C = 1); minv: (N, 6) float64 DEVICE tensor, the dst -> src matrices; -> (N,C,Hd,Wd) f32."""
    N, C, Hs, Ws = src.shape
    assert src.is_contiguous() and src.dtype in (torch.float32, torch.float64)
    assert minv.dtype == torch.float64 and minv.is_contiguous() and tuple(minv.shape) == (N, 6) and minv.is_cuda
    Hd, Wd = out_hw
    out = torch.empty((N, C, Hd, Wd), dtype=torch.float32, device=src.device)
    b = (ctypes.c_float * 4)(*([float(v) for v in border] + [0.0] * 4)[:4])
    check(lib().flair_warp_affine_cubic(ptr(src), int(src.dtype == torch.float64), N, C, Hs, Ws, ptr(minv), Hd, Wd, b,
                                        int(pre), int(post), ptr(out), stream()), "flair_warp_affine_cubic")
    return out


def face_mask_blur(parse_idx, N, H, W, lut, kern, *, repeats=2, edge=10, div=255.0):
    """lut[parse_idx] -> repeats x separable float64 Gaussian blur (reflect-101) -> edge zeroed, / div
    (flair_face_mask_blur).  parse_idx: (N*H*W,) int32; lut, kern: float64 DEVICE tensors -> (N,1,H,W) float64."""
    assert parse_idx.dtype == torch.int32 and parse_idx.is_contiguous() and parse_idx.numel() == N * H * W
    assert lut.dtype == torch.float64 and kern.dtype == torch.float64 and lut.is_cuda and kern.is_cuda
    tmp = torch.empty((N, 1, H, W), dtype=torch.float64, device=parse_idx.device)
    mask = torch.empty_like(tmp)
    check(lib().flair_face_mask_blur(ptr(parse_idx), N, H, W, ptr(lut), lut.numel(), ptr(kern), kern.numel(), repeats, edge,
                                     ctypes.c_double(div), ptr(tmp), ptr(mask), stream()), "flair_face_mask_blur")
    return mask


def face_blend(x0, face, mask):
    """x0 * (1 - mask) + face * mask; mask (N,1,H,W) f32 broadcast over the channels."""
    N, C, H, W = x0.shape
    assert x0.dtype == face.dtype == mask.dtype == torch.float32 and x0.is_contiguous() and face.is_contiguous()
    assert mask.is_contiguous() and tuple(mask.shape) == (N, 1, H, W) and face.shape == x0.shape
    out = torch.empty_like(x0)
    check(lib().flair_face_blend(ptr(x0), ptr(face), ptr(mask), N, C, H, W, ptr(out), stream()), "flair_face_blend")
    return out
