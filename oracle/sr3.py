"""CPU restatement of the SR3-style video UNet of the bicubic tasks (x8 / x16).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows guided_diffusion/sr3.py (UNet and
its blocks, :45-525) and the blocks it borrows from guided_diffusion/unet.py (ResBlock with a
(3,1,1) kernel :113-254, TemporalAttention :664-758, BasicVSRPP with its own flow computation
:313-595).  Shipped configuration only: spatial_attn=False, with_noise_level_emb=True,
use_affine_level=False.  Parameter names reproduce the reference state-dict.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .thirdparty import SPyNet
from .unet import (BasicVSRPP, ResBlock, TemporalAttention, Wrapped, _zero, group_norm_over_clip,
                   per_frame)


class PositionalEncoding(nn.Module):
    """sr3.py:45-60: [sin | cos](level * 1e4^(-i/count))."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, noise_level):
        count = self.dim // 2
        step = torch.arange(count, dtype=noise_level.dtype) / count
        enc = noise_level.unsqueeze(1) * torch.exp(-math.log(1e4) * step.unsqueeze(0))
        return torch.cat([torch.sin(enc), torch.cos(enc)], dim=-1)


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class FeatureWiseAffine(nn.Module):
    """sr3.py:63-83 without the affine level: x + Linear(emb) per (frame, channel)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.noise_func = nn.Sequential(nn.Linear(in_channels, out_channels))

    def forward(self, x, emb):
        b, t = x.shape[:2]
        return x + self.noise_func(emb).to(x.dtype).view(b, t, -1, 1, 1)


class Block(nn.Module):
    """sr3.py:113-126: GroupNorm over the clip -> Swish -> conv3x3 per frame."""

    def __init__(self, dim, dim_out, groups=32):
        super().__init__()
        self.block = nn.Sequential(Wrapped(nn.GroupNorm(groups, dim)), Swish(), nn.Identity(),
                                   Wrapped(nn.Conv2d(dim, dim_out, 3, padding=1)))

    def forward(self, x):
        h = group_norm_over_clip(x, self.block[0].wrapped_module)
        h = h * torch.sigmoid(h)
        return per_frame(h, self.block[3].wrapped_module)


class ResnetBlock(nn.Module):
    """sr3.py:129-161."""

    def __init__(self, dim, dim_out, emb_dim, norm_groups=32):
        super().__init__()
        self.noise_func = FeatureWiseAffine(emb_dim, dim_out)
        self.block1 = Block(dim, dim_out, groups=norm_groups)
        self.block2 = Block(dim_out, dim_out, groups=norm_groups)
        self.res_conv = Wrapped(nn.Conv2d(dim, dim_out, 1)) if dim != dim_out else nn.Identity()

    def forward(self, x, emb):
        h = self.block2(self.noise_func(self.block1(x), emb))
        if isinstance(self.res_conv, nn.Identity):
            return h + x
        return h + per_frame(x, self.res_conv.wrapped_module)


class TemporalWrapper2(nn.Module):
    """sr3.py:203-226: (1 - s) * x + s * module(x), s = sigmoid(Linear(SiLU(emb))) per (frame, channel)."""

    def __init__(self, module, dim, time_emb_dim):
        super().__init__()
        self.wrapped_module = module
        self.emb_layers = nn.Sequential(nn.SiLU(), _zero(nn.Linear(time_emb_dim, dim)))

    def gate(self, x, emb):
        b, n, c = x.shape[:3]
        return torch.sigmoid(self.emb_layers(emb).view(b, n, c, 1, 1).to(x.dtype))

    def forward(self, x, emb, out):
        s = self.gate(x, emb)
        return (1 - s) * x + s * out


class FlowVSRPP(BasicVSRPP):
    """unet.py:313-595: BasicVSR++ that resizes the low-quality clip to its own resolution
    (antialiased bilinear) and runs the shared SPyNet itself."""

    def __init__(self, mid_channels, max_residue_magnitude, shared_spynet):
        super().__init__(mid_channels, max_residue_magnitude)
        self.spynet = shared_spynet

    @torch.no_grad()
    def compute_flow(self, lqs):
        lqs = ((lqs + 1) / 2).clamp(0, 1)
        n, t, c, h, w = lqs.shape
        a = lqs[:, :-1].reshape(-1, c, h, w)
        b = lqs[:, 1:].reshape(-1, c, h, w)
        return self.spynet(b, a).view(n, t - 1, 2, h, w), self.spynet(a, b).view(n, t - 1, 2, h, w)

    def forward(self, hidden, lqs, weight=None):
        if lqs.shape[-2:] != hidden.shape[-2:]:
            lqs = per_frame(lqs, lambda z: F.interpolate(z, size=tuple(hidden.shape[-2:]), mode="bilinear",
                                                         align_corners=False, antialias=True))
        assert lqs.shape[-1] >= 64 and lqs.shape[-2] >= 64
        flows_forward, flows_backward = self.compute_flow(lqs)
        return super().forward(hidden, flows_forward, flows_backward, weight)


class ResnetBlocWithAttn(nn.Module):
    """sr3.py:229-314."""

    def __init__(self, dim, dim_out, *, emb_dim, norm_groups, conv_3d, temporal_attn, num_frames, head_dim,
                 vsrpp, shared_spynet):
        super().__init__()
        self.res_block = ResnetBlock(dim, dim_out, emb_dim, norm_groups=norm_groups)
        if conv_3d:
            self.conv_3d = TemporalWrapper2(
                ResBlock(dim_out, emb_dim, dims=3, use_scale_shift_norm=False, kernel_size=(3, 1, 1),
                         padding=(1, 0, 0)), dim_out, emb_dim)
        if temporal_attn:
            self.temp_attn = TemporalWrapper2(
                TemporalAttention(dim_out, num_frames=num_frames, num_heads=8, num_head_channels=head_dim),
                dim_out, emb_dim)
        if vsrpp:
            self.vsrpp = TemporalWrapper2(FlowVSRPP(dim_out, 5, shared_spynet), dim_out, emb_dim)

    def forward(self, x, lqs, emb, cross_frame_enabled=True, vsrpp_weights=None):
        x = self.res_block(x, emb)
        if hasattr(self, "conv_3d") and cross_frame_enabled:
            x = self.conv_3d(x, emb, self.conv_3d.wrapped_module(x, emb))
        if hasattr(self, "temp_attn") and cross_frame_enabled:
            x = self.temp_attn(x, emb, self.temp_attn.wrapped_module(x))
        if hasattr(self, "vsrpp") and cross_frame_enabled:
            x = self.vsrpp(x, emb, self.vsrpp.wrapped_module(x, lqs, weight=vsrpp_weights))
        return x


class Downsample(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv2d(dim, dim, 3, 2, 1)

    def forward(self, x):
        return self.conv(x)


class Upsample(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv2d(dim, dim, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode="nearest"))


class UNet(nn.Module):
    """sr3.py:317-525."""

    takes_noise_level = True

    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                 channel_mults=(1, 2, 4, 8, 8), attn_res=(8,), vsrpp_res=(64,), spatial_attn=False,
                 temporal_attn=False, res_blocks=3, dropout=0, with_noise_level_emb=True, image_size=128,
                 dtype=torch.float32, cross_frame_module=False, use_checkpoint=False, num_frames=5, head_dim=32):
        super().__init__()
        assert not spatial_attn and with_noise_level_emb, "oracle covers the shipped sr3 configuration"
        shared = SPyNet() if len(vsrpp_res) > 0 else None
        emb = inner_channel
        self.noise_level_mlp = nn.Sequential(PositionalEncoding(inner_channel),
                                             nn.Linear(inner_channel, inner_channel * 4), Swish(),
                                             nn.Linear(inner_channel * 4, inner_channel))

        def blk(cin, cout, res):
            return ResnetBlocWithAttn(
                cin, cout, emb_dim=emb, norm_groups=norm_groups, conv_3d=cross_frame_module,
                temporal_attn=(res in attn_res and temporal_attn and cross_frame_module), num_frames=num_frames,
                head_dim=head_dim, vsrpp=(res in vsrpp_res and cross_frame_module), shared_spynet=shared)

        pre, res = inner_channel, image_size
        feats = [pre]
        downs = [Wrapped(nn.Conv2d(in_channel, inner_channel, 3, padding=1))]
        n = len(channel_mults)
        for i in range(n):
            ch = inner_channel * channel_mults[i]
            for _ in range(res_blocks):
                downs.append(blk(pre, ch, res))
                feats.append(ch)
                pre = ch
            if i != n - 1:
                downs.append(Wrapped(Downsample(pre)))
                feats.append(pre)
                res //= 2
        self.downs = nn.ModuleList(downs)
        mid_kw = dict(emb_dim=emb, norm_groups=norm_groups, conv_3d=cross_frame_module,
                      temporal_attn=temporal_attn and cross_frame_module, num_frames=num_frames,
                      head_dim=head_dim, vsrpp=False, shared_spynet=None)
        self.mid = nn.ModuleList([ResnetBlocWithAttn(pre, pre, **mid_kw), ResnetBlocWithAttn(pre, pre, **mid_kw)])
        ups = []
        for i in reversed(range(n)):
            ch = inner_channel * channel_mults[i]
            for _ in range(res_blocks + 1):
                ups.append(blk(pre + feats.pop(), ch, res))
                pre = ch
            if i >= 1:
                ups.append(Wrapped(Upsample(pre)))
                res *= 2
        self.ups = nn.ModuleList(ups)
        self.final_conv = Block(pre, out_channel if out_channel is not None else in_channel, groups=norm_groups)

    def forward(self, x, timesteps, low_res_input=None, rnn_input=None, num_frames=None,
                enable_cross_frames=True, vsrpp_weights=None, **kwargs):
        if rnn_input is None:
            rnn_input = low_res_input
        x = x.reshape(-1, num_frames, *x.shape[1:])
        x = torch.cat((low_res_input, x), dim=2)
        t = self.noise_level_mlp(timesteps)
        feats = []
        for layer in self.downs:
            if isinstance(layer, ResnetBlocWithAttn):
                x = layer(x, rnn_input, t, enable_cross_frames, vsrpp_weights)
            else:
                x = per_frame(x, layer.wrapped_module)
            feats.append(x)
        for layer in self.mid:
            x = layer(x, rnn_input, t, enable_cross_frames, vsrpp_weights)
        for layer in self.ups:
            if isinstance(layer, ResnetBlocWithAttn):
                x = layer(torch.cat((x, feats.pop()), dim=2), rnn_input, t, enable_cross_frames, vsrpp_weights)
            else:
                x = per_frame(x, layer.wrapped_module)
        y = self.final_conv(x)
        return y.reshape(-1, *y.shape[2:])
