"""Per-kernel parity: HIP path (through the C ABI) vs the CPU oracle / plain torch fp32.

Run on the GPU box with ``pytest -m gpu``.  Integer-free floating point work: the
tolerance per dtype is stated in tests/util.py.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close, from_clip, rb, to_clip

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def _ops():
    from flair_amd import ops
    return ops


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # T, H, W, segs, cout, kernel, act, nres
    (2, 16, 16, [64], 64, (1, 3, 3), 0, 0),
    (3, 20, 12, [64, 32], 128, (1, 3, 3), 2, 1),
    (4, 8, 8, [32], 96, (3, 3, 3), 0, 2),
    (1, 9, 7, [32], 16, (1, 7, 7), 1, 0),
    (2, 16, 16, [128], 432, (1, 3, 3), 0, 0),
    (5, 6, 6, [64], 8, (1, 1, 1), 3, 0),
    (1, 64, 64, [64, 64, 64, 32], 64, (1, 3, 3), 2, 0),
    (16, 4, 4, [512], 512, (3, 3, 3), 0, 1),
    # halo kernel (3x3 spatial taps, W % 32 == 0): 8 / 4 / 2 rows per workgroup, 2-D and 3-D
    (16, 64, 64, [64], 64, (1, 3, 3), 0, 1),
    (2, 128, 128, [64, 64], 64, (1, 3, 3), 2, 0),
    (1, 64, 32, [128], 432, (1, 3, 3), 0, 0),
    (1, 30, 32, [64, 64, 64, 32], 64, (1, 3, 3), 2, 2),
    (5, 16, 32, [64], 128, (3, 3, 3), 0, 1),
    (16, 36, 64, [32], 64, (3, 3, 3), 3, 0),
    (16, 72, 64, [32], 64, (3, 3, 3), 1, 1),
    (2, 136, 128, [64], 64, (1, 3, 3), 2, 1),
    # K-split halo kernel: launches of exactly 256 workgroups (one 256^2 / 128^2 frame)
    (1, 256, 256, [64, 32], 64, (1, 3, 3), 2, 2),
    (4, 64, 64, [64], 128, (3, 3, 3), 0, 1),
    (1, 250, 256, [64], 64, (1, 3, 3), 3, 1),       # last row of tiles hangs over the image
    (1, 125, 128, [32], 128, (1, 3, 3), 0, 0),
    # one-round launches of 4-row tiles (the 128^2 level, c = 128): the K-split kernel / conv_frame_ks_kernel
    (1, 128, 128, [128], 128, (1, 3, 3), 2, 1),
    (1, 128, 128, [128, 128, 128, 32], 128, (1, 3, 3), 2, 0),
    (1, 128, 128, [64], 72, (1, 3, 3), 1, 2),
    (2, 64, 128, [32], 128, (1, 3, 3), 0, 0),
    # conv_frame_kernel (round 4: one 256-tile round, Cout <= 64): one / two / seven K chunks, padded couts, each residual count
    (1, 256, 256, [32], 64, (1, 3, 3), 0, 1),
    (1, 256, 256, [64], 24, (1, 3, 3), 1, 1),
    (1, 256, 256, [64, 64, 64, 32], 64, (1, 3, 3), 2, 0),
    (4, 128, 128, [32, 32], 40, (1, 3, 3), 1, 2),
    # persistent LDS-DMA kernel (bf16; f32 takes the halo kernel): >= 192 tiles of 16 rows x 32 px x 64 couts --
    # one tile per workgroup, 2 and 1.5 tiles per workgroup, three temporal taps with clip edges, 432 couts, 4 segments
    (4, 128, 128, [64], 128, (1, 3, 3), 0, 1),
    (8, 128, 128, [64], 128, (1, 3, 3), 3, 2),
    (3, 256, 256, [32], 64, (1, 3, 3), 2, 0),
    (6, 64, 128, [32, 32], 128, (3, 3, 3), 1, 1),
    (6, 64, 128, [64], 128, (3, 3, 3), 0, 0),
    (3, 256, 256, [32], 8, (1, 3, 3), 0, 1),        # 8 couts: half of every 16-cout store group is padding
    (6, 128, 128, [96], 64, (3, 3, 3), 0, 1),
    (1, 128, 128, [128], 432, (1, 3, 3), 0, 0),
    (2, 128, 256, [64, 32, 64, 32], 64, (1, 3, 3), 2, 2),
    # deep-K convolutions on few pixels: 128 x 128 tiles with split-K (16x16 / 8x8 levels)
    (8, 16, 16, [256], 256, (3, 3, 3), 3, 1),
    (16, 8, 8, [256, 256], 384, (1, 3, 3), 0, 2),
])
def test_conv(dev, dtype, case):
    ops = _ops()
    T, H, W, segs, cout, k, act, nres = case
    g = torch.Generator().manual_seed(T * 1000 + H * 10 + cout)
    cin = sum(segs)
    x = rb(torch.randn(T, cin, H, W, generator=g), dtype)
    fan = cin * k[0] * k[1] * k[2]
    w = rb(torch.randn(cout, cin, *k, generator=g) / math.sqrt(fan), dtype)
    b = torch.randn(cout, generator=g) * 0.1
    res = [rb(torch.randn(T, cout, H, W, generator=g), dtype) for _ in range(nres)]
    # reference (fp32, cpu)
    if k[0] == 1:
        ref = F.conv2d(x, w[:, :, 0], b, padding=(k[1] // 2, k[2] // 2))
    else:
        ref = F.conv3d(x.permute(1, 0, 2, 3)[None], w, b, padding=tuple(v // 2 for v in k))[0].permute(1, 0, 2, 3)
    ref = {0: lambda v: v, 1: F.relu, 2: lambda v: F.leaky_relu(v, 0.1), 3: F.silu}[act](ref)
    for r in res:
        ref = ref + r
    ref = ref * 0.5
    xs, o = [], 0
    for c in segs:
        xs.append(to_clip(x[:, o:o + c], dtype, dev))
        o += c
    wp = ops.pack_conv_weight(w, [(c, c) for c in segs], dtype).to(dev)
    rs = [to_clip(r, dtype, dev) for r in res] + [None, None]
    y = ops.conv(xs, wp, b.to(dev), cout, k, act=act, res0=rs[0], res1=rs[1], out_scale=0.5)
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, f"conv {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # T, H, W, cin, cout, kernel, nres: per-(frame, cout) bias (the ResBlock's emb term, unet_new.py:258-264,321) on the
    # LDS-DMA kernel (persistent and one-tile forms, with and without a residual, padded cout tile), the K-split and the igemm kernels
    (4, 128, 128, 64, 128, (1, 3, 3), 0), (4, 128, 128, 64, 128, (3, 3, 3), 1), (3, 256, 256, 32, 72, (1, 3, 3), 1),
    (1, 256, 256, 64, 64, (1, 3, 3), 0), (1, 128, 128, 128, 128, (1, 3, 3), 1), (5, 16, 16, 64, 64, (1, 3, 3), 0),
])
def test_conv_frame_bias(dev, dtype, case):
    ops = _ops()
    T, H, W, cin, cout, k, nres = case
    g = torch.Generator().manual_seed(T * 31 + H + cout)
    x = rb(torch.randn(T, cin, H, W, generator=g), dtype)
    w = rb(torch.randn(cout, cin, *k, generator=g) / math.sqrt(cin * k[0] * 9), dtype)
    b = torch.randn(cout, generator=g) * 0.1
    fb = torch.randn(T, cout, generator=g)
    res = [rb(torch.randn(T, cout, H, W, generator=g), dtype) for _ in range(nres)]
    if k[0] == 1:
        ref = F.conv2d(x, w[:, :, 0], b, padding=1)
    else:
        ref = F.conv3d(x.permute(1, 0, 2, 3)[None], w, b, padding=1)[0].permute(1, 0, 2, 3)
    ref = ref + fb[:, :, None, None]
    for r in res:
        ref = ref + r
    wp = ops.pack_conv_weight(w, [(cin, cin)], dtype).to(dev)
    y = ops.conv([to_clip(x, dtype, dev)], wp, b.to(dev), cout, k, frame_bias=fb.to(dev),
                 res0=to_clip(res[0], dtype, dev) if res else None)
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, f"conv frame_bias {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # T, H, W, segs (None: no stage A), c_mid, coutB, actA, actB, nres
    (1, 256, 256, [64], 64, 64, 2, 2, 0),              # conv_offset[2]+[4] at the 256^2 level (8 x 32 tiles)
    (1, 128, 128, [128], 128, 128, 1, 0, 2),           # ResidualBlockNoBN at the 128^2 level (8 x 8 tiles)
    (1, 64, 64, [64], 64, 432, 2, 0, 0),               # conv_offset[4]+[6]: wide second stage
    (1, 128, 128, [128], 128, 432, 2, 0, 0),
    (1, 256, 256, None, 64, 432, 0, 0, 0),             # resident-input wide convolution alone
    (1, 128, 128, None, 128, 432, 0, 0, 0),
    (3, 32, 32, [64, 64, 32], 64, 64, 2, 1, 1),        # several frames, three input segments, small frame (8 x 8 tiles)
    (2, 20, 40, [64], 64, 64, 3, 2, 2),                # tiles hang over the image bottom; W % 32 != 0
    (16, 64, 64, [64], 64, 64, 1, 0, 1),               # clip-level (reconstruction trunk), >= 192 wide tiles
    (1, 250, 256, [64, 64], 64, 64, 2, 0, 1),          # partial last row of tiles at 256 wide
    (1, 24, 16, None, 64, 72, 0, 2, 1),                # CoutB not a multiple of 64
    (2, 16, 64, [64], 64, 64, 2, 0, 2),                # the c = 64 pair with both residual inputs, two frames
    (1, 8, 32, [64], 64, 64, 0, 1, 1),                 # one tile: every border of the halo and of the intermediate is padding
    (1, 256, 256, [64], 64, 64, 1, 2, 2),              # ResidualBlockNoBN at the 256^2 level
])
def test_conv_chain(dev, dtype, case):
    """Two fused 3x3 convolutions (flair_conv_chain) against two plain torch convolutions, the intermediate
    rounded to the element type exactly where the two-launch path rounds it."""
    ops = _ops()
    T, H, W, segs, cm, coutB, actA, actB, nres = case
    g = torch.Generator().manual_seed(T * 977 + H * 13 + coutB)
    acts = {0: lambda v: v, 1: F.relu, 2: lambda v: F.leaky_relu(v, 0.1), 3: F.silu}
    cin = sum(segs) if segs else cm
    x = rb(torch.randn(T, cin, H, W, generator=g), dtype)
    wB = rb(torch.randn(coutB, cm, 3, 3, generator=g) / math.sqrt(9 * cm), dtype)
    bB = torch.randn(coutB, generator=g) * 0.1
    res = [rb(torch.randn(T, coutB, H, W, generator=g), dtype) for _ in range(nres)]
    if segs:
        wA = rb(torch.randn(cm, cin, 3, 3, generator=g) / math.sqrt(9 * cin), dtype)
        bA = torch.randn(cm, generator=g) * 0.1
        mid = rb(acts[actA](F.conv2d(x, wA, bA, padding=1)), dtype)
    else:
        wA = bA = None
        mid = x
    ref = acts[actB](F.conv2d(mid, wB, bB, padding=1))
    for r in res:
        ref = ref + r
    ref = ref * 0.5
    xs, o = [], 0
    for c in (segs or [cm]):
        xs.append(to_clip(x[:, o:o + c], dtype, dev))
        o += c
    wAp = ops.pack_conv_weight(wA[:, :, None], [(c, c) for c in segs], dtype).to(dev) if segs else None
    wBp = ops.pack_conv_weight(wB[:, :, None], [(cm, cm)], dtype).to(dev)
    rs = [to_clip(r, dtype, dev) for r in res] + [None, None]
    y = ops.conv_chain(xs, wAp, bA.to(dev) if segs else None, actA, wBp, bB.to(dev), actB, cm, coutB,
                       res0=rs[0], res1=rs[1], out_scale=0.5)
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, f"conv_chain {case}")


def test_conv_cases_cover_every_kernel_variant():
    """The geometries above dispatch to the 10 conv kernel variants in use (flair_conv_variant; variants 10 / 11, the 4-row
    one-tile LDS-DMA forms, are selected with FLAIR_CONV_DMA_FRAME=2 / 3 only: the default environment is assumed here)."""
    import os
    if os.environ.get("FLAIR_CONV_DMA_FRAME", "1") != "1":
        pytest.skip("FLAIR_CONV_DMA_FRAME overrides the default dispatch")
    ops = _ops()
    geo = [(2, 16, 16, [64], 64, (1, 3, 3)), (16, 64, 64, [64], 64, (1, 3, 3)), (2, 128, 128, [64, 64], 64, (1, 3, 3)),
           (1, 64, 32, [128], 432, (1, 3, 3)), (1, 30, 32, [64, 64, 64, 32], 64, (1, 3, 3)),
           (16, 72, 64, [32], 64, (3, 3, 3)), (2, 136, 128, [64], 64, (1, 3, 3)),
           (1, 256, 256, [64, 32], 64, (1, 3, 3)), (4, 64, 64, [64], 128, (3, 3, 3)),
           (16, 256, 256, [64], 64, (1, 1, 1)), (16, 128, 128, [128], 128, (1, 1, 1)), (16, 4, 4, [512], 512, (3, 3, 3)),
           (4, 128, 128, [64], 128, (1, 3, 3)), (16, 264, 256, [64], 64, (1, 3, 3)),
           (1, 250, 256, [64], 64, (1, 3, 3)), (1, 125, 128, [32], 128, (1, 3, 3))]
    seen = {ops.conv_variant(T, H, W, segs, cout, k) for T, H, W, segs, cout, k in geo}
    assert seen == set(range(10)), seen


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [
    # T, H, W, c0, c1, film, resample, fps
    (4, 16, 16, 64, 0, False, 0, None),
    (3, 12, 20, 128, 64, True, 0, None),
    (2, 16, 16, 64, 0, True, 1, None),
    (2, 8, 8, 256, 0, False, 2, None),
    (4, 8, 8, 96, 0, True, 0, 1),
    (16, 4, 4, 1024, 0, True, 0, None),
    (3, 4, 4, 1024, 1024, True, 0, None),          # 2048 channels: more 16-byte slots than 256 threads in f32
    # one-launch small-tensor kernel (a workgroup per group, <= 128 KB per group; the first case is just above): two segments, per-frame stats
    (16, 32, 32, 256, 0, True, 0, None),
    (16, 16, 16, 256, 256, True, 0, None),
    (4, 8, 8, 256, 0, True, 0, 1),
    (8, 16, 16, 512, 0, False, 0, None),
])
def test_group_norm(dev, dtype, case):
    ops = _ops()
    T, H, W, c0, c1, film, resample, fps = case
    C = c0 + c1
    g = torch.Generator().manual_seed(7 + C)
    x = rb(torch.randn(T, C, H, W, generator=g) * 1.7 + 0.3, dtype)
    gamma = torch.randn(C, generator=g)
    beta = torch.randn(C, generator=g)
    emb = torch.randn(T, 2 * C + (5 if C < 512 else 8), generator=g) * 0.5 if film else None   # rows (un)aligned to 16 B
    if fps == 1:
        n = F.group_norm(x, 32, gamma, beta, 1e-5)
    else:
        n = F.group_norm(x.permute(1, 0, 2, 3)[None], 32, gamma, beta, 1e-5)[0].permute(1, 0, 2, 3)
    if film:
        n = n * (1 + emb[:, :C, None, None]) + emb[:, C:2 * C, None, None]
    ref = F.silu(n)
    raw_ref = x
    if resample == 1:
        ref, raw_ref = F.avg_pool2d(ref, 2), F.avg_pool2d(x, 2)
    elif resample == 2:
        ref, raw_ref = F.interpolate(ref, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    xa = to_clip(x[:, :c0], dtype, dev)
    xb = to_clip(x[:, c0:], dtype, dev) if c1 else None
    res = ops.group_norm(xa, gamma.to(dev), beta.to(dev), x1=xb, act=ops.ACT_SILU,
                         film=emb.to(dev) if film else None, frames_per_stat=fps,
                         resample=resample, want_raw=bool(resample))
    y, raw = res if resample else (res, None)
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, f"gn {case}", scale=4.0)
    if resample:
        assert_close(from_clip(raw), raw_ref, dtype, f"gn raw {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("new_order", [False, True])
@pytest.mark.parametrize("case", [(2, 4, 4, 2), (3, 8, 8, 4), (2, 16, 16, 4), (1, 32, 32, 1), (1, 12, 12, 2),
                                  (16, 16, 16, 4),      # the L = 256 blocks of config 2 (64-query workgroups)
                                  (16, 32, 32, 4),      # L = 1024 (512x512 clips): 128-query workgroups, 16 KV tiles
                                  (5, 20, 20, 13)])     # L = 400: last KV tile partly masked, 128-query workgroups
def test_qkv_attention(dev, dtype, new_order, case):
    from oracle.unet import qkv_attention_legacy, qkv_attention_new
    ops = _ops()
    Fr, H, W, heads = case
    C = heads * 64
    g = torch.Generator().manual_seed(11)
    qkv = rb(torch.randn(Fr, 3 * C, H * W, generator=g) * 1.5, dtype)
    ref = (qkv_attention_new if new_order else qkv_attention_legacy)(qkv, heads).reshape(Fr, C, H, W)
    y = ops.qkv_attention(to_clip(qkv.reshape(Fr, 3 * C, H, W), dtype, dev), heads, new_order=new_order)
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, f"attn {case}", scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(4, 4, 4, 128), (7, 8, 8, 64), (2, 2, 2, 256)])
def test_temporal_attention(dev, dtype, case):
    from oracle.thirdparty import flash_attn_func
    ops = _ops()
    T, H, W, C = case
    heads = C // 64
    g = torch.Generator().manual_seed(5)
    qkv = rb(torch.randn(T, 3 * C, H, W, generator=g), dtype)
    kpos = torch.randn(4, C, generator=g) * 0.3
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    idx = (torch.arange(T).view(T, 1) + torch.tensor([-2, -1, 1, 2]).view(1, 4)).clamp(0, T - 1)
    kw = k[idx] + kpos.view(1, 4, C, 1, 1)             # T,4,C,H,W
    vw = v[idx]
    def tok(z):  # -> (T*H*W, n, heads, 64)
        n = z.shape[1]
        return z.permute(0, 3, 4, 1, 2).reshape(T * H * W, n, heads, 64)
    qq = tok(q[:, None])
    if dtype == torch.float32:   # reference rounds through fp16 (nn.py:370-386)
        o = flash_attn_func(qq.half(), tok(kw).half(), tok(vw).half()).float()
    else:
        o = flash_attn_func(qq, tok(kw), tok(vw))
    ref = o.reshape(T, H, W, C).permute(0, 3, 1, 2)
    y = ops.temporal_attention(to_clip(qkv, dtype, dev), kpos.to(dev), 5, round_fp16=(dtype == torch.float32))
    torch.cuda.synchronize()
    # f32 path reproduces the fp16 rounding of the reference: 1 fp16 ulp (2^-10) slack
    err = (from_clip(y) - ref).abs().max().item()
    bound = (2e-3 if dtype == torch.float32 else 1.6e-2) * ref.abs().max().item() + 1e-3
    assert err <= bound, (err, bound)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("border", [False, True])
def test_flow_warp(dev, dtype, border):
    from oracle.thirdparty import flow_warp
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = rb(torch.randn(3, 64, 12, 10, generator=g), dtype)
    flow = torch.randn(3, 12, 10, 2, generator=g) * 3.0
    ref = flow_warp(x, flow, padding_mode="border" if border else "zeros")
    y = ops.flow_warp(to_clip(x, dtype, dev), flow.to(dev), border=border)
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, "flow_warp", scale=4.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("second", [False, True])
def test_vsrpp_warp2(dev, dtype, second):
    """The two warps of one propagation step in one launch (flair_vsrpp_warp2; unet_new.py:706,719 = mmedit flow_warp with
    zeros padding) against the oracle, with flows that push whole regions outside the frame and strided output views."""
    from oracle.thirdparty import flow_warp
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    H, W, C = 20, 24, 64
    prop = rb(torch.randn(1, C, H, W, generator=g), dtype)
    feat2 = rb(torch.randn(1, C, H, W, generator=g), dtype)
    f1 = torch.randn(1, H, W, 2, generator=g) * 4.0
    f2 = torch.randn(1, H, W, 2, generator=g) * 9.0
    f1[:, :3] += 30.0                                  # rows whose four corners all fall outside
    tdt = torch.bfloat16 if dtype == torch.bfloat16 else torch.float32
    buf = torch.zeros(1, H, W, 3 * C, device=dev, dtype=tdt)      # cond1 | (gap) | cond2 as views of one wider clip tensor
    c1, c2 = buf[..., :C], buf[..., 2 * C:]
    ops.vsrpp_warp2(to_clip(prop, dtype, dev), to_clip(feat2, dtype, dev) if second else None, f1.to(dev),
                    f2.to(dev) if second else None, c1, c2 if second else None)
    torch.cuda.synchronize()
    assert_close(from_clip(c1), flow_warp(prop, f1, padding_mode="zeros"), dtype, "vsrpp_warp2 cond1", scale=4.0)
    if second:
        assert_close(from_clip(c2), flow_warp(feat2, f2, padding_mode="zeros"), dtype, "vsrpp_warp2 cond2", scale=4.0)
    assert buf[..., C:2 * C].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_vsrpp_prep_and_warp2_agree_bit_for_bit(dev, dtype):
    """The first forward of a clip composes the second-order flow and warps in flair_vsrpp_prep, every later one warps with
    the cached flow in flair_vsrpp_warp2: a replayed hipGraph equals the eager first call only if the two kernels round
    identically (they once differed by an FMA contraction in the bilinear weights)."""
    ops = _ops()
    tdt = torch.bfloat16 if dtype == torch.bfloat16 else torch.float32
    g = torch.Generator().manual_seed(5)
    for H, W, C in ((32, 32, 64), (20, 24, 128)):
        prop = torch.randn(1, H, W, C, generator=g).to(dev).to(tdt)
        feat2 = torch.randn(1, H, W, C, generator=g).to(dev).to(tdt)
        f1 = (torch.randn(1, H, W, 2, generator=g) * 3).to(dev)
        fprev = (torch.randn(1, H, W, 2, generator=g) * 3).to(dev)
        c1, c2, f2 = torch.empty_like(prop), torch.empty_like(prop), torch.empty_like(f1)
        pad = torch.zeros(1, H, W, 32, device=dev, dtype=tdt)
        ops.vsrpp_prep(prop, feat2, f1, fprev, c1, c2, f2, pad)
        d1, d2 = torch.empty_like(prop), torch.empty_like(prop)
        ops.vsrpp_warp2(prop, feat2, f1, f2, d1, d2)
        torch.cuda.synchronize()
        assert torch.equal(c1, d1) and torch.equal(c2, d2)


def test_flow_compose(dev):
    from oracle.thirdparty import flow_warp
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    f1 = torch.randn(2, 9, 11, 2, generator=g) * 2
    f2 = torch.randn(2, 9, 11, 2, generator=g) * 2
    ref = f1 + flow_warp(f2.permute(0, 3, 1, 2), f1).permute(0, 2, 3, 1)
    y = ops.flow_compose(f1.to(dev), f2.to(dev))
    torch.cuda.synchronize()
    assert_close(y.cpu(), ref, torch.float32, "flow_compose", scale=8.0)


@pytest.mark.parametrize("mode,size", [(0, (24, 40)), (0, (10, 7)), (1, (32, 24)), (2, (8, 6)), (2, (40, 30)), (3, (8, 6))])
def test_resize(dev, mode, size):
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 16, 12, generator=g)
    if mode == 0:
        ref = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    elif mode == 1:
        ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    elif mode == 2:
        ref = F.interpolate(x, size=size, mode="bicubic")
    else:
        ref = F.avg_pool2d(x, 2, 2)
    y = ops.resize(to_clip(x, torch.float32, dev, pad_to=4), size, mode, channels=3)
    torch.cuda.synchronize()
    assert_close(from_clip(y, 3), ref, torch.float32, f"resize {mode}", scale=8.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,hw", [(64, (12, 10)), (128, (12, 10)), (64, (20, 27)), (128, (9, 23))])
def test_dcn(dev, dtype, c, hw):
    from oracle.thirdparty import deform_conv2d
    ops = _ops()
    g = torch.Generator().manual_seed(21 + c)
    H, W, G = hw[0], hw[1], 16
    x = rb(torch.randn(1, 2 * c, H, W, generator=g), dtype)
    raw = rb(torch.randn(1, 27 * G, H, W, generator=g), dtype)
    f1 = torch.randn(1, H, W, 2, generator=g) * 2
    f2 = torch.randn(1, H, W, 2, generator=g) * 2
    w = rb(torch.randn(c, 2 * c, 3, 3, generator=g) / math.sqrt(18 * c), dtype)
    b = torch.randn(c, generator=g) * 0.1
    o1, o2, mask = raw.chunk(3, dim=1)
    offset = 10 * torch.tanh(torch.cat((o1, o2), dim=1))
    off1, off2 = offset.chunk(2, dim=1)
    off1 = off1 + f1.permute(0, 3, 1, 2).flip(1).repeat(1, off1.shape[1] // 2, 1, 1)
    off2 = off2 + f2.permute(0, 3, 1, 2).flip(1).repeat(1, off2.shape[1] // 2, 1, 1)
    ref = deform_conv2d(x, torch.cat([off1, off2], 1), w, b, (1, 1), (1, 1), (1, 1), torch.sigmoid(mask))
    wp = ops.pack_conv_weight(w, [(2 * c, 2 * c)], dtype).to(dev)
    raw_tap_major = raw[:, ops.dcn_raw_permutation(G)]     # the layout the fused kernel reads
    y = ops.dcn_align(to_clip(x[:, :c], dtype, dev), to_clip(x[:, c:], dtype, dev),
                      to_clip(raw_tap_major, dtype, dev), f1.to(dev), f2.to(dev), wp, b.to(dev), c)
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, f"dcn c={c}", scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,hw", [(64, (32, 32)), (128, (16, 40)), (128, (128, 128)), (64, (256, 256))])
def test_dcn_activated_raw(dev, dtype, c, hw):
    """flair_dcn_align with raw_activated=1: residues / masks arrive finished (the producing convolution's
    FLAIR_ACT_DCN_OFFSETS epilogue), rounded to the element type like any stored activation."""
    from oracle.thirdparty import deform_conv2d
    ops = _ops()
    g = torch.Generator().manual_seed(33 + c)
    H, W, G = hw[0], hw[1], 16
    x = rb(torch.randn(1, 2 * c, H, W, generator=g), dtype)
    raw = torch.randn(1, 27 * G, H, W, generator=g)
    f1 = torch.randn(1, H, W, 2, generator=g) * 2
    f2 = torch.randn(1, H, W, 2, generator=g) * 2
    w = rb(torch.randn(c, 2 * c, 3, 3, generator=g) / math.sqrt(18 * c), dtype)
    b = torch.randn(c, generator=g) * 0.1
    o1, o2, mask = raw.chunk(3, dim=1)
    act = torch.cat([rb(10 * torch.tanh(torch.cat((o1, o2), dim=1)), dtype), rb(torch.sigmoid(mask), dtype)], dim=1)
    off1, off2 = act[:, :18 * G].chunk(2, dim=1)
    off1 = off1 + f1.permute(0, 3, 1, 2).flip(1).repeat(1, off1.shape[1] // 2, 1, 1)
    off2 = off2 + f2.permute(0, 3, 1, 2).flip(1).repeat(1, off2.shape[1] // 2, 1, 1)
    ref = deform_conv2d(x, torch.cat([off1, off2], 1), w, b, (1, 1), (1, 1), (1, 1), act[:, 18 * G:])
    wp = ops.pack_conv_weight(w, [(2 * c, 2 * c)], dtype).to(dev)
    y = ops.dcn_align(to_clip(x[:, :c], dtype, dev), to_clip(x[:, c:], dtype, dev),
                      to_clip(act[:, ops.dcn_raw_permutation(G)], dtype, dev), f1.to(dev), f2.to(dev), wp, b.to(dev), c,
                      raw_activated=True)
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, f"dcn activated c={c}", scale=2.0)


@pytest.mark.parametrize("c,hw", [(64, (16, 32)), (128, (8, 16)), (64, (11, 13))])
def test_dcn_vs_reference_source(dev, c, hw):
    """flair_dcn_align (f32, finished residues / masks) against oracle/dcn_ref.py, the literal restatement of the
    reference's own DCNv2 source (dcn/src/deform_conv_cuda_kernel.cu:468-497,571-633), on offsets that put sampling
    positions in (-1, 0), at and beyond H - 1 / W - 1, exactly on integers, exactly at -1 / H and far outside."""
    from oracle import dcn_ref
    ops = _ops()
    dtype = torch.float32
    H, W, G = hw[0], hw[1], 16
    g = torch.Generator().manual_seed(77 + c)
    x = torch.randn(1, 2 * c, H, W, generator=g)
    w = torch.randn(c, 2 * c, 3, 3, generator=g) / math.sqrt(18 * c)
    b = torch.randn(c, generator=g) * 0.1
    offset = torch.randn(1, 2 * G * 9, H, W, generator=g) * 3
    mask = torch.rand(1, G * 9, H, W, generator=g)
    off = offset.view(1, G, 9, 2, H, W)
    hh = torch.arange(H, dtype=torch.float32).view(1, H, 1)
    ww = torch.arange(W, dtype=torch.float32).view(1, 1, W)
    targets = [(-0.5, 0.25), (-1.0, 2.0), (-0.999, -0.001), (H - 1.0, W - 1.0), (H - 0.5, W - 0.25),
               (float(H), 1.0), (2.0, 3.0), (H - 1.25, -1.0), (-3.0, 1.5)]
    for k in range(9):
        i, j = divmod(k, 3)
        off[:, k, k, 0] = (targets[k][0] - (hh - 1 + i)).expand(1, H, W)
        off[:, k, k, 1] = (targets[k][1] - (ww - 1 + j)).expand(1, H, W)
    off[:, G - 1] = torch.round(off[:, G - 1])
    ref = torch.from_numpy(dcn_ref.modulated_deform_conv_forward(x.numpy(), w.numpy(), b.numpy(), offset.numpy(),
                                                                 mask.numpy(), deformable_group=G))
    act = torch.cat([offset, mask], dim=1)                      # the reference's channel order: (o1 | o2 | mask)
    wp = ops.pack_conv_weight(w, [(2 * c, 2 * c)], dtype).to(dev)
    y = ops.dcn_align(to_clip(x[:, :c], dtype, dev), to_clip(x[:, c:], dtype, dev),
                      to_clip(act[:, ops.dcn_raw_permutation(G)], dtype, dev), None, None, wp, b.to(dev), c,
                      raw_activated=True)
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, f"dcn vs reference source c={c}", scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("path", ["halo", "igemm", "chain"])
def test_dcn_offset_activation_epilogue(dev, dtype, path):
    """FLAIR_ACT_DCN_OFFSETS in the epilogue of the offset convolution (tap-major output: per 48 channels
    32 residues -> 10*tanh, 16 masks -> sigmoid), on the three kernels that can produce it."""
    ops = _ops()
    G, cin = 16, 64
    H, W = (16, 32) if path != "igemm" else (12, 20)
    g = torch.Generator().manual_seed(5)
    x = rb(torch.randn(1, cin, H, W, generator=g), dtype)
    w = rb(torch.randn(27 * G, cin, 3, 3, generator=g) / math.sqrt(9 * cin) * 3, dtype)
    b = torch.randn(27 * G, generator=g) * 0.3
    pre = F.conv2d(x, w, b, padding=1)
    ch = torch.arange(27 * G)
    residue = (ch % (3 * G)) < 2 * G
    ref = torch.where(residue.view(1, -1, 1, 1), 10 * torch.tanh(pre), torch.sigmoid(pre))
    wp = ops.pack_conv_weight(w[:, :, None], [(cin, cin)], dtype).to(dev)
    xc = to_clip(x, dtype, dev)
    if path == "chain":
        y = ops.conv_chain(xc, None, None, 0, wp, b.to(dev), ops.ACT_DCN_OFFSETS, cin, 27 * G, act_param=10.0,
                           act_period=3 * G)
    else:
        y = ops.conv(xc, wp, b.to(dev), 27 * G, (1, 3, 3), act=ops.ACT_DCN_OFFSETS, act_param=10.0, act_period=3 * G)
    torch.cuda.synchronize()
    # tanh / sigmoid through __expf + v_rcp: 1e-6 relative; the bound is on max|ref| = 10
    assert_close(from_clip(y), ref, dtype, f"dcn offset activation ({path})", scale=2.0)


@pytest.mark.parametrize("cout,act,shape", [(432, 4, (1, 16, 32)), (432, 4, (2, 24, 64)), (64, 2, (1, 8, 32)), (200, 0, (3, 8, 96)),
                                            (8, 1, (1, 16, 32)), (72, 3, (1, 32, 32)), (432, 4, (1, 128, 128)), (432, 4, (1, 256, 256))])
def test_conv_resident_input(dev, cout, act, shape):
    """conv_resident_kernel (round 4: flair_conv_chain without a first stage on bf16 c = 64 inputs -- the c -> 27*G offset
    convolution): halo resident in LDS, weight ring by LDS-DMA, epilogue of block b beside the MFMAs of block b + 1.
    Odd block counts, a partial last block, one-block outputs, every activation class, several frames; against F.conv2d."""
    ops = _ops()
    dtype = torch.bfloat16
    T, H, W = shape
    cin, G = 64, 16
    g = torch.Generator().manual_seed(100 + cout + H)
    x = rb(torch.randn(T, cin, H, W, generator=g), dtype)
    w = rb(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin) * (3 if act == 4 else 1), dtype)
    b = torch.randn(cout, generator=g) * 0.3
    pre = F.conv2d(x, w, b, padding=1)
    if act == 4:
        residue = (torch.arange(cout) % (3 * G)) < 2 * G
        ref = torch.where(residue.view(1, -1, 1, 1), 10 * torch.tanh(pre), torch.sigmoid(pre))
    else:
        ref = {0: lambda v: v, 1: torch.relu, 2: lambda v: F.leaky_relu(v, 0.1), 3: F.silu}[act](pre)
    ref = ref * 0.75
    wp = ops.pack_conv_weight(w[:, :, None], [(cin, cin)], dtype).to(dev)
    y = torch.full((T, H, W, cout + 8), 7.0, dtype=dtype, device=dev)          # a wider buffer: the pad channels must stay untouched
    ops.conv_chain(to_clip(x, dtype, dev), None, None, 0, wp, b.to(dev), act, cin, cout, out=y, out_scale=0.75,
                   act_param=10.0 if act == 4 else 0.0, act_period=3 * G if act == 4 else 0)
    torch.cuda.synchronize()
    assert torch.all(y[..., cout:] == 7.0)
    assert_close(from_clip(y[..., :cout]), ref, dtype, f"resident-input conv cout={cout} act={act}", scale=2.0)


def test_embedding_linear_layout(dev):
    from oracle.unet import timestep_embedding
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    t = torch.tensor([0., 1., 37., 999., 500.5])
    ref = timestep_embedding(t, 128)
    y = ops.timestep_embedding(t.to(dev), 128)
    assert_close(y.cpu(), ref, torch.float32, "timestep_embedding", scale=50.0)
    x = torch.randn(16, 512, generator=g)
    w = torch.randn(1000, 512, generator=g) / 22
    b = torch.randn(1000, generator=g)
    ref = F.linear(F.silu(x), w, b)
    y = ops.linear(x.to(dev), w.to(dev), b.to(dev), act_in=ops.ACT_SILU)
    assert_close(y.cpu(), ref, torch.float32, "linear")
    img = torch.randn(4, 3, 10, 6, generator=g)
    for dt in DTYPES:
        clip = torch.zeros(4, 10, 6, 32, dtype=dt, device=dev)
        ops.nchw_to_clip(img.to(dev), clip, coff=3)
        back = ops.clip_to_nchw(clip, 3, coff=3)
        assert_close(back.cpu(), rb(img, dt), dt, "layout")
        assert clip[..., :3].abs().sum().item() == 0 and clip[..., 6:].abs().sum().item() == 0


def test_error_paths(dev):
    from flair_amd import _lib, ops
    x = torch.zeros(1, 4, 4, 24, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(8, 9, 24, dtype=torch.bfloat16, device=dev)
    with pytest.raises(_lib.FlairHipError, match="multiple of 32"):
        ops.conv(x, w, None, 8, (1, 3, 3))
    with pytest.raises(_lib.FlairHipError):
        ops.conv(torch.zeros(1, 4, 4, 32), torch.zeros(8, 9, 32), None, 8, (1, 3, 3))  # CPU tensors


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_flash_attn_wrapper_calling_convention(dev, dtype):
    """nn.flash_attn_wrapper(q, k, v, dropout) on (B, L, heads, D) tensors (nn.py:370-386) against exact attention on the
    fp16-rounded inputs (flash-attn itself is absent: published definition softmax(q k^T / sqrt(D)) v, parity unpinned)."""
    from flair_amd.guided_diffusion import nn as fnn
    g = torch.Generator().manual_seed(12)
    B, L, Hh, D = 6, 5, 2, 64
    q, k, v = (torch.randn(B, L, Hh, D, generator=g) for _ in range(3))
    qh, kh, vh = (t.half().float() for t in (q, k, v))
    if dtype == torch.bfloat16:
        qh, kh, vh = (t.bfloat16().float() for t in (qh, kh, vh))
    att = torch.softmax(torch.einsum("blhd,bmhd->bhlm", qh, kh) / math.sqrt(D), dim=-1)
    ref = torch.einsum("bhlm,bmhd->blhd", att, vh)
    got = fnn.flash_attn_wrapper(q.to(dev).to(dtype), k.to(dev).to(dtype), v.to(dev).to(dtype), 0.0)
    torch.cuda.synchronize()
    assert got.shape == (B, L, Hh, D) and got.dtype == dtype
    assert_close(got.float().cpu(), ref, dtype, "flash_attn_wrapper")
    with pytest.raises(NotImplementedError):
        fnn.flash_attn_wrapper(q.to(dev), k.to(dev), v.to(dev), 0.1)


@pytest.mark.gpu
def test_calibration_launches():
    """flair_probe_matrix_rate / flair_probe_stream_rate (measurement-only entries of the C ABI): they run, report their work, the copy copies."""
    ops = _ops()
    flop = ops.probe_matrix_rate(10, 1)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert flop == cus * 4 * 10 * 16 * 2.0 * 32 * 32 * 16
    src = torch.randint(0, 255, (1 << 20,), dtype=torch.uint8, device="cuda")
    dst = torch.zeros_like(src)
    assert ops.probe_stream_rate(src, dst, 2) == 2 << 20
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    assert ops.probe_stream_rate(src, dst, 0) == 1 << 20
    with pytest.raises(Exception):
        ops.probe_matrix_rate(0, 1)
