"""Restatements of third-party callees that the reference imports but does not vendor.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: none of these
packages is importable in the build container, so the
functions below follow the published algorithm of the pinned versions
(requirements.txt of the reference):

  * mmedit==0.12.0   flow_warp, ResidualBlocksWithInputConv, ResidualBlockNoBN,
                     SPyNet / SPyNetBasicModule
                     (call sites: guided_diffusion/unet_new.py:21-25,659-667,
                     706-719,985,1305-1307)
  * mmcv-full==1.4.8 ModulatedDeformConv2d (parameter container only),
                     constant_init (unet_new.py:26-27,835,857,872)
  * torchvision==0.15.2  ops.deform_conv2d (unet_new.py:889-898)
  * flash-attn       flash_attn_func (guided_diffusion/nn.py:15,378-383)
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- mmedit
def flow_warp(x, flow, interpolation="bilinear", padding_mode="zeros", align_corners=True):
    """Warp ``x`` (n,c,h,w) by ``flow`` (n,h,w,2; [...,0]=dx, [...,1]=dy).

    Sampling position for output pixel (i,j) is (j+dx, i+dy) in pixel units; it is
    normalised to [-1,1] with ``max(size-1,1)`` and handed to grid_sample.
    """
    n, c, h, w = x.shape
    ys = torch.arange(h, dtype=x.dtype, device=x.device).view(1, h, 1)
    xs = torch.arange(w, dtype=x.dtype, device=x.device).view(1, 1, w)
    px = xs + flow[..., 0]
    py = ys + flow[..., 1]
    gx = 2.0 * px / max(w - 1, 1) - 1.0
    gy = 2.0 * py / max(h - 1, 1) - 1.0
    grid = torch.stack((gx, gy), dim=3)
    return F.grid_sample(x, grid, mode=interpolation, padding_mode=padding_mode,
                         align_corners=align_corners)


class ResidualBlockNoBN(nn.Module):
    """x + conv2(relu(conv1(x))) with 3x3 convs (res_scale = 1)."""

    def __init__(self, mid_channels=64):
        super().__init__()
        self.conv1 = nn.Conv2d(mid_channels, mid_channels, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(mid_channels, mid_channels, 3, 1, 1, bias=True)

    def forward(self, x):
        return x + self.conv2(F.relu(self.conv1(x)))


class ResidualBlocksWithInputConv(nn.Module):
    """conv3x3(in->out) + LeakyReLU(0.1) + ``num_blocks`` ResidualBlockNoBN.

    State-dict names follow mmedit: ``main.0`` (conv), ``main.2.<i>.conv1/conv2``.
    """

    def __init__(self, in_channels, out_channels=64, num_blocks=30):
        super().__init__()
        self.main = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, 3, 1, 1, bias=True),
            nn.LeakyReLU(negative_slope=0.1, inplace=False),
            nn.Sequential(*[ResidualBlockNoBN(out_channels) for _ in range(num_blocks)]),
        )

    def forward(self, feat):
        return self.main(feat)


class _ConvAct(nn.Module):
    """mmcv ConvModule reduced to what SPyNet uses: ``.conv`` + optional ReLU."""

    def __init__(self, cin, cout, relu):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=7, stride=1, padding=3)
        self.relu = relu

    def forward(self, x):
        x = self.conv(x)
        return F.relu(x) if self.relu else x


class SPyNetBasicModule(nn.Module):
    """Five 7x7 convs 8->32->64->32->16->2, ReLU between (none after the last)."""

    def __init__(self):
        super().__init__()
        chans = [8, 32, 64, 32, 16, 2]
        self.basic_module = nn.Sequential(
            *[_ConvAct(chans[i], chans[i + 1], relu=(i < 4)) for i in range(5)]
        )

    def forward(self, tensor_input):
        return self.basic_module(tensor_input)


class SPyNet(nn.Module):
    """Six-level coarse-to-fine optical flow (Ranjan & Black), mmedit flavour."""

    def __init__(self, pretrained=None):
        super().__init__()
        self.basic_module = nn.ModuleList([SPyNetBasicModule() for _ in range(6)])
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def compute_flow(self, ref, supp):
        n, _, h, w = ref.shape
        pyr_r = [(ref - self.mean) / self.std]
        pyr_s = [(supp - self.mean) / self.std]
        for _ in range(5):
            pyr_r.append(F.avg_pool2d(pyr_r[-1], 2, 2, count_include_pad=False))
            pyr_s.append(F.avg_pool2d(pyr_s[-1], 2, 2, count_include_pad=False))
        pyr_r.reverse()
        pyr_s.reverse()
        flow = ref.new_zeros(n, 2, h // 32, w // 32)
        for level in range(6):
            if level == 0:
                up = flow
            else:
                up = 2.0 * F.interpolate(flow, scale_factor=2, mode="bilinear",
                                         align_corners=True)
            warped = flow_warp(pyr_s[level], up.permute(0, 2, 3, 1), padding_mode="border")
            flow = up + self.basic_module[level](torch.cat([pyr_r[level], warped, up], 1))
        return flow

    def forward(self, ref, supp):
        h, w = ref.shape[2:4]
        w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
        h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
        ref = F.interpolate(ref, size=(h_up, w_up), mode="bilinear", align_corners=False)
        supp = F.interpolate(supp, size=(h_up, w_up), mode="bilinear", align_corners=False)
        flow = F.interpolate(self.compute_flow(ref, supp), size=(h, w), mode="bilinear",
                             align_corners=False)
        flow = flow.clone()
        flow[:, 0] *= float(w) / float(w_up)
        flow[:, 1] *= float(h) / float(h_up)
        return flow


class PixelShufflePack(nn.Module):  # imported by the reference, never constructed on the path
    def __init__(self, *a, **k):
        super().__init__()


# ----------------------------------------------------------------------------- mmcv
def constant_init(module, val, bias=0):
    if getattr(module, "weight", None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


class ModulatedDeformConv2d(nn.Module):
    """Parameter container with mmcv's attribute names (weight, bias, stride, ...)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, groups=1, deform_groups=1, bias=True):
        super().__init__()
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, k
        self.stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
        self.padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.dilation = (dilation, dilation) if isinstance(dilation, int) else tuple(dilation)
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *k))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        n = in_channels * k[0] * k[1]
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.zero_()


# ---------------------------------------------------------------------- torchvision
def deform_conv2d(x, offset, weight, bias=None, stride=(1, 1), padding=(0, 0),
                  dilation=(1, 1), mask=None):
    """Modulated deformable convolution v2 (torchvision.ops.deform_conv2d semantics).

    offset: (n, 2*G*kh*kw, ho, wo) with channel 2*(g*kh*kw+k) = dy, +1 = dx;
    mask:   (n, G*kh*kw, ho, wo); bilinear sampling, zero outside the image.
    out[n,co] = bias[co] + sum_{ci,k} W[co,ci,k] * mask[g(ci),k] * sample(x[ci], p0+pk+d).

    Pinned to source the reference itself holds: the same operator exists there as
    guided_diffusion/dcn/src/deform_conv_cuda_kernel.cu:468-497 (dmcn_im2col_bilinear), :571-633
    (modulated_deformable_im2col_gpu_kernel: offset channel 2*(g*kh*kw+k) = row displacement, +1 = column
    displacement, mask channel g*kh*kw+k, the `h_im > -1 && h_im < height` window) and deform_conv_cuda.cpp:540-560
    (im2col + addmm + bias).  oracle/dcn_ref.py restates those lines literally in numpy (loops, no grid_sample) and
    tests/test_dcn_ref_cpu.py holds this function to it (f64: 1e-11, incl. positions in (-1, 0), >= H-1, exactly
    integer, exactly -1 / H and far outside).
    """
    def _pair(v):
        return (v, v) if isinstance(v, int) else tuple(v)

    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    n, cin, h, w = x.shape
    cout, cin_g, kh, kw = weight.shape
    assert cin_g == cin, "grouped weights are not used on the FLAIR path"
    taps = kh * kw
    groups = offset.shape[1] // (2 * taps)
    cpg = cin // groups
    ho = (h + 2 * padding[0] - dilation[0] * (kh - 1) - 1) // stride[0] + 1
    wo = (w + 2 * padding[1] - dilation[1] * (kw - 1) - 1) // stride[1] + 1
    base_y = (torch.arange(ho, dtype=x.dtype, device=x.device) * stride[0] - padding[0]).view(1, ho, 1)
    base_x = (torch.arange(wo, dtype=x.dtype, device=x.device) * stride[1] - padding[1]).view(1, 1, wo)
    xg = x.reshape(n * groups, cpg, h, w)
    off = offset.reshape(n, groups, taps, 2, ho, wo)
    cols = []
    for k in range(taps):
        ki, kj = divmod(k, kw)
        py = base_y + ki * dilation[0] + off[:, :, k, 0].reshape(n * groups, ho, wo)
        px = base_x + kj * dilation[1] + off[:, :, k, 1].reshape(n * groups, ho, wo)
        gx = 2.0 * px / max(w - 1, 1) - 1.0
        gy = 2.0 * py / max(h - 1, 1) - 1.0
        s = F.grid_sample(xg, torch.stack((gx, gy), dim=-1), mode="bilinear",
                          padding_mode="zeros", align_corners=True)
        s = s.reshape(n, groups, cpg, ho, wo)
        if mask is not None:
            s = s * mask.reshape(n, groups, taps, ho, wo)[:, :, k].unsqueeze(2)
        cols.append(s.reshape(n, cin, ho, wo))
    col = torch.stack(cols, dim=2)  # n, cin, taps, ho, wo
    out = torch.einsum("oik,nikhw->nohw", weight.reshape(cout, cin, taps), col)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


# ----------------------------------------------------------------------- flash-attn
def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False):
    """Exact attention for (batch, seqlen, heads, dim) tensors, fp32 accumulate."""
    assert dropout_p == 0.0 and not causal
    d = q.shape[-1]
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(d)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * scale
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhqk,bkhd->bqhd", p, v.float())
    return o.to(q.dtype)
