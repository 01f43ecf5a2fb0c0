"""CPU restatement of the FLAIR video UNet (gaussian / jpeg tasks) in plain PyTorch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
guided_diffusion/unet_new.py of the reference; every class cites the lines it
restates.  Tensors are (B, T, C, H, W) between blocks, exactly as in the
reference; parameter names reproduce the reference state-dict (SURVEY.md
Appendix B) so weights can be exchanged with fixtures and with the HIP model.

Unlike the reference this module builds the deformable-alignment branch on any
device (the reference only creates it when CUDA is present, unet_new.py:650) --
the oracle always models the GPU-shaped network.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .thirdparty import (ModulatedDeformConv2d, ResidualBlocksWithInputConv, SPyNet,
                         deform_conv2d, flash_attn_func, flow_warp)


# ------------------------------------------------------------------ small helpers
class Wrapped(nn.Module):
    """Holds a child under the attribute name the reference uses (nn.py:340-367)."""

    def __init__(self, module):
        super().__init__()
        self.wrapped_module = module


def _zero(module):
    for p in module.parameters():
        nn.init.zeros_(p)
    return module


def timestep_embedding(timesteps, dim, max_period=10000):
    """[cos | sin](t * max_period^(-i/half)) -- nn_new.py:103-121."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None].to(timesteps.device)
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm_over_clip(x, gn):
    """GroupNorm32 with statistics over (C/32, T, H, W) -- nn_new.py:17-19 behind
    LazyReshaper3D (nn.py:359-367).  x: (B,T,C,H,W); computed in fp32."""
    y = F.group_norm(x.permute(0, 2, 1, 3, 4).float(), gn.num_groups, gn.weight.float(),
                     gn.bias.float(), gn.eps)
    return y.permute(0, 2, 1, 3, 4).to(x.dtype)


def per_frame(x, fn):
    """Apply a 4-D op frame by frame -- LazyReshaper2D (nn.py:350-356)."""
    b, t = x.shape[:2]
    y = fn(x.reshape(b * t, *x.shape[2:]))
    return y.reshape(b, t, *y.shape[1:])


def over_clip(x, fn):
    """Apply a 5-D (B,C,T,H,W) op -- LazyReshaper3D."""
    return fn(x.permute(0, 2, 1, 3, 4)).permute(0, 2, 1, 3, 4)


# ------------------------------------------------------------------------ ResBlock
class ResBlock(nn.Module):
    """unet_new.py:198-329.  dims=2: per-frame 3x3 convs; dims=3: 3x3x3 over (T,H,W).
    GroupNorm always spans the whole clip.  up/down: nearest x2 / avg-pool 2 applied
    to both branches before the first conv (unet_new.py:310-315)."""

    def __init__(self, channels, emb_channels, out_channels=None, dims=2,
                 use_scale_shift_norm=True, up=False, down=False, kernel_size=3, padding=1):
        """kernel_size / padding: the guided_diffusion/unet.py:113-254 variant used by sr3
        (e.g. a (3,1,1) temporal kernel); 3 / 1 reproduces unet_new.py."""
        super().__init__()
        out_channels = out_channels or channels
        self.channels, self.out_channels, self.dims = channels, out_channels, dims
        self.use_scale_shift_norm = use_scale_shift_norm
        self.up, self.down = up, down
        conv = nn.Conv2d if dims == 2 else nn.Conv3d
        self.in_layers = nn.Sequential(
            Wrapped(nn.GroupNorm(32, channels)), nn.SiLU(),
            Wrapped(conv(channels, out_channels, kernel_size, padding=padding)))
        self.emb_layers = nn.Sequential(
            nn.SiLU(), nn.Linear(emb_channels, 2 * out_channels if use_scale_shift_norm
                                 else out_channels))
        self.out_layers = nn.Sequential(
            Wrapped(nn.GroupNorm(32, out_channels)), nn.SiLU(), nn.Dropout(0.0),
            _zero(Wrapped(conv(out_channels, out_channels, kernel_size, padding=padding))))
        if out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = Wrapped(conv(channels, out_channels, 1))

    def _conv(self, wrapped, x):
        if self.dims == 2:
            return per_frame(x, wrapped.wrapped_module)
        return over_clip(x, wrapped.wrapped_module)

    def _resample(self, x):
        if self.up:
            return per_frame(x, lambda z: F.interpolate(z, scale_factor=2, mode="nearest"))
        if self.down:
            return per_frame(x, lambda z: F.avg_pool2d(z, 2, 2))
        return x

    def forward(self, x, emb):
        t = x.shape[1]
        h = F.silu(group_norm_over_clip(x, self.in_layers[0].wrapped_module))
        h = self._resample(h)
        x = self._resample(x)
        h = self._conv(self.in_layers[2], h)
        e = self.emb_layers(emb).to(h.dtype).reshape(-1, t, self.emb_layers[1].out_features, 1, 1)
        if self.use_scale_shift_norm:
            scale, shift = e.chunk(2, dim=2)
            h = group_norm_over_clip(h, self.out_layers[0].wrapped_module) * (1 + scale) + shift
            h = F.silu(h)
        else:
            h = F.silu(group_norm_over_clip(h + e, self.out_layers[0].wrapped_module))
        h = self._conv(self.out_layers[3], h)
        if isinstance(self.skip_connection, nn.Identity):
            return x + h
        return self._conv(self.skip_connection, x) + h


# ----------------------------------------------------------------------- attention
def qkv_attention_legacy(qkv, n_heads):
    """unet_new.py:540-570: heads first, then q|k|v inside each head block."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).to(w.dtype)
    return torch.einsum("bts,bcs->bct", w, v).reshape(bs, -1, length)


def qkv_attention_new(qkv, n_heads):
    """unet_new.py:573-605: q|k|v first, then heads."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.chunk(3, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", (q * scale).reshape(bs * n_heads, ch, length),
                     (k * scale).reshape(bs * n_heads, ch, length))
    w = torch.softmax(w.float(), dim=-1).to(w.dtype)
    return torch.einsum("bts,bcs->bct", w, v.reshape(bs * n_heads, ch, length)).reshape(
        bs, -1, length)


class AttentionBlock(nn.Module):
    """unet_new.py:332-377 (and :380-429 when ``bottleneck``: adds SiLU->Linear(512,512)
    of the timestep embedding to the attention output before proj_out)."""

    def __init__(self, channels, num_head_channels=64, num_heads=1, new_order=False,
                 bottleneck=False):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.new_order, self.bottleneck = new_order, bottleneck
        if bottleneck:
            self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(512, 512))
        self.norm = Wrapped(nn.GroupNorm(32, channels))
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = _zero(nn.Conv1d(channels, channels, 1))

    def forward(self, x, emb=None):
        b, n, c, hh, ww = x.shape
        y = group_norm_over_clip(x, self.norm.wrapped_module).reshape(b * n, c, hh * ww)
        qkv = self.qkv(y)
        a = (qkv_attention_new if self.new_order else qkv_attention_legacy)(qkv, self.num_heads)
        if self.bottleneck:
            a = a + self.emb_layers(emb).to(a.dtype).unsqueeze(-1)
        return x + self.proj_out(a).reshape(b, n, c, hh, ww)


class TemporalAttention(nn.Module):
    """unet_new.py:432-517: every pixel attends, per head, from its own frame (query,
    positional code of offset 0) to the ``num_frames-1`` neighbouring frames
    (replicate-padded at the clip ends) with sinusoidal codes of their offsets."""

    def __init__(self, channels, num_frames=5, num_heads=1, num_head_channels=64):
        super().__init__()
        self.channels, self.num_frames = channels, num_frames
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.q_linear = nn.Linear(channels, channels)
        self.k_linear = nn.Linear(channels, channels)
        self.v_linear = nn.Linear(channels, channels)
        self.proj = _zero(Wrapped(nn.Conv2d(channels, channels, 1)))
        self.norm = Wrapped(nn.GroupNorm(32, channels))
        offs = torch.arange(num_frames, dtype=torch.long) - num_frames // 2
        pe = timestep_embedding(offs, channels)
        mid = num_frames // 2
        self.t_mid = pe[mid:mid + 1]
        self.t_rest = pe[torch.arange(num_frames) != mid]

    def forward(self, hid):
        b, t, c, hh, ww = hid.shape
        half = self.num_frames // 2
        x = group_norm_over_clip(hid, self.norm.wrapped_module)
        idx = (torch.arange(t).view(t, 1) + torch.arange(-half, half + 1).view(1, -1)).clamp(0, t - 1)
        win = x[:, idx]                                   # b, t, f, c, h, w
        win = win.permute(0, 1, 4, 5, 2, 3).reshape(b * t * hh * ww, self.num_frames, c)
        keep = torch.arange(self.num_frames) != half
        q = self.q_linear(win[:, half:half + 1] + self.t_mid.to(win.dtype))
        kv_in = win[:, keep]
        k = self.k_linear(kv_in + self.t_rest.to(win.dtype))
        v = self.v_linear(kv_in)
        nh = self.num_heads

        def heads(z):
            return z.reshape(z.shape[0], z.shape[1], nh, c // nh)

        # the reference always rounds q/k/v (and the result) through fp16 here,
        # whatever the model dtype (flash_attn_wrapper, nn.py:370-386)
        a = flash_attn_func(heads(q).half(), heads(k).half(), heads(v).half(), 0.0).to(q.dtype)
        a = a.reshape(b, t, hh, ww, c).permute(0, 1, 4, 2, 3)
        return per_frame(a, self.proj.wrapped_module) + hid


# ---------------------------------------------------------------------- BasicVSR++
class SecondOrderDeformableAlignment(ModulatedDeformConv2d):
    """unet_new.py:835-898: offsets = 10*tanh(conv stack) + flow (dy,dx order, first
    half of the groups follows flow_1, second half flow_2); mask = sigmoid."""

    def __init__(self, *args, max_residue_magnitude=10, **kwargs):
        super().__init__(*args, **kwargs)
        self.max_residue_magnitude = max_residue_magnitude
        c = self.out_channels
        self.conv_offset = nn.Sequential(
            nn.Conv2d(3 * c + 4, c, 3, 1, 1), nn.LeakyReLU(0.1),
            nn.Conv2d(c, c, 3, 1, 1), nn.LeakyReLU(0.1),
            nn.Conv2d(c, c, 3, 1, 1), nn.LeakyReLU(0.1),
            nn.Conv2d(c, 27 * self.deform_groups, 3, 1, 1))
        _zero(self.conv_offset[-1])

    def forward(self, x, extra_feat, flow_1, flow_2):
        out = self.conv_offset(torch.cat([extra_feat, flow_1, flow_2], dim=1))
        o1, o2, mask = out.chunk(3, dim=1)
        offset = self.max_residue_magnitude * torch.tanh(torch.cat((o1, o2), dim=1))
        off1, off2 = offset.chunk(2, dim=1)
        off1 = off1 + flow_1.flip(1).repeat(1, off1.shape[1] // 2, 1, 1)
        off2 = off2 + flow_2.flip(1).repeat(1, off2.shape[1] // 2, 1, 1)
        return deform_conv2d(x, torch.cat([off1, off2], dim=1), self.weight, self.bias,
                             self.stride, self.padding, self.dilation, torch.sigmoid(mask))


class BasicVSRPP(nn.Module):
    """unet_new.py:608-832: one backward then one forward second-order propagation
    over the frames of a clip, then a per-frame reconstruction conv stack, a
    zero-initialised 1x1 conv and a residual onto the input."""

    def __init__(self, mid_channels=64, max_residue_magnitude=10):
        super().__init__()
        self.mid_channels = mid_channels
        self.deform_align = nn.ModuleDict()
        self.backbone = nn.ModuleDict()
        for i, name in enumerate(["backward_1", "forward_1"]):
            self.deform_align[name] = SecondOrderDeformableAlignment(
                2 * mid_channels, mid_channels, 3, padding=1, deform_groups=16,
                max_residue_magnitude=max_residue_magnitude)
            self.backbone[name] = ResidualBlocksWithInputConv((2 + i) * mid_channels, mid_channels, 1)
        self.reconstruction = ResidualBlocksWithInputConv(3 * mid_channels, mid_channels, 1)
        self.conv_last = _zero(nn.Conv2d(mid_channels, mid_channels, 1, 1))

    def _propagate(self, feats, flows, name, weight):
        n, tm1, _, h, w = flows.shape
        t = tm1 + 1
        order = list(range(t))
        flow_idx = list(range(-1, tm1))
        if "backward" in name:
            order = order[::-1]
            flow_idx = order
        spatial = feats["spatial"]
        prop = flows.new_zeros(n, self.mid_channels, h, w, dtype=spatial[0].dtype)
        outs = []
        for i, idx in enumerate(order):
            cur = spatial[idx]
            if i > 0:
                flow_n1 = flows[:, flow_idx[i]].to(cur.dtype)
                cond_n1 = flow_warp(prop, flow_n1.permute(0, 2, 3, 1))
                feat_n2 = torch.zeros_like(prop)
                flow_n2 = torch.zeros_like(flow_n1)
                cond_n2 = torch.zeros_like(cond_n1)
                if i > 1:
                    feat_n2 = outs[-2]
                    flow_n2 = flows[:, flow_idx[i - 1]].to(cur.dtype)
                    flow_n2 = flow_n1 + flow_warp(flow_n2, flow_n1.permute(0, 2, 3, 1))
                    cond_n2 = flow_warp(feat_n2, flow_n2.permute(0, 2, 3, 1))
                cond = torch.cat([cond_n1, cur, cond_n2], dim=1)
                prop = self.deform_align[name](torch.cat([prop, feat_n2], dim=1), cond,
                                               flow_n1, flow_n2)
            others = [feats[k][idx] for k in feats if k not in ("spatial", name)]
            prop = prop + self.backbone[name](torch.cat([cur] + others + [prop], dim=1))
            # NB: the reference multiplies in place *after* storing the tensor
            # (unet_new.py:738-739), so the stored feature is the weighted one too.
            prop = prop * weight[:, idx]
            outs.append(prop)
        if "backward" in name:
            outs = outs[::-1]
        feats[name] = outs
        return feats

    def forward(self, hidden, flows_forward, flows_backward, weight=None):
        n, t, c, h, w = hidden.shape
        if weight is None:
            weight = torch.ones(n, t, 1, 1, 1, device=hidden.device)
        elif isinstance(weight, float):
            weight = torch.ones(n, t, 1, 1, 1, device=hidden.device) * weight
        elif weight.shape[-2] != h or weight.shape[-1] != w:
            weight = per_frame(weight, lambda z: F.interpolate(z, size=(h, w), mode="nearest"))
        feats = {"spatial": [hidden[:, i] for i in range(t)]}
        for name in ("backward_1", "forward_1"):
            if "backward" in name:
                flows = flows_backward
            elif flows_forward is not None:
                flows = flows_forward
            else:
                flows = flows_backward.flip(1)
            feats = self._propagate(feats, flows, name, weight)
        recons = []
        for i in range(t):
            hr = torch.cat([feats["spatial"][i], feats["backward_1"][i], feats["forward_1"][i]], dim=1)
            recons.append(self.reconstruction(hr))
        recons = torch.stack(recons, dim=1)
        return per_frame(recons, self.conv_last) + hidden


# ----------------------------------------------------------------------- the UNet
class Stage(nn.Sequential):
    """TimestepEmbedSequential (unet_new.py:106-133): routes emb / flows by layer type."""

    def forward(self, x, emb, flows, vsrpp_weights, enable_cross_frames=True):
        for layer in self:
            inner = layer.wrapped_module if isinstance(layer, Wrapped) else layer
            temporal = isinstance(layer, Wrapped) and not isinstance(inner, nn.Conv2d)
            if temporal and not enable_cross_frames:
                continue
            if isinstance(inner, ResBlock):
                x = inner(x, emb)
            elif isinstance(inner, AttentionBlock):
                x = inner(x, emb)
            elif isinstance(inner, TemporalAttention):
                x = inner(x)
            elif isinstance(inner, BasicVSRPP):
                fwd, bwd = flows[x.shape[-1]]
                x = inner(x, fwd, bwd, vsrpp_weights)
            elif isinstance(inner, nn.Conv2d):
                x = per_frame(x, inner)
            else:
                raise TypeError(type(inner))
        return x


class UNetModel(nn.Module):
    """unet_new.py:901-1362 for the shipped gaussian/jpeg configuration family
    (resblock_updown=True, use_scale_shift_norm=True, temporal_block=True)."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, rnn_resolutions, dropout=0, channel_mult=(1, 2, 4, 8),
                 conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, temporal_block=False):
        super().__init__()
        assert dims == 2 and num_classes is None and resblock_updown, \
            "oracle covers the shipped FLAIR configuration family only"
        self.image_size, self.model_channels = image_size, model_channels
        self.need_flows_res = [image_size // s for s in rnn_resolutions]
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.spynet = SPyNet()

        def res(cin, cout, d=2, **kw):
            return ResBlock(cin, ted, cout, dims=d, use_scale_shift_norm=use_scale_shift_norm, **kw)

        def attn(ch, bottleneck=False):
            return AttentionBlock(ch, num_head_channels, num_heads, use_new_attention_order, bottleneck)

        def level_layers(cin, cout, ds):
            layers = [res(cin, cout)]
            if temporal_block:
                layers.append(Wrapped(res(cout, cout, 3)))
            if ds in attention_resolutions:
                layers.append(attn(cout))
                if temporal_block:
                    layers.append(Wrapped(TemporalAttention(cout, 5, num_heads, num_head_channels)))
            if ds in rnn_resolutions and temporal_block:
                layers.append(Wrapped(BasicVSRPP(cout)))
            return layers

        ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([Stage(Wrapped(nn.Conv2d(in_channels, ch, 3, padding=1)))])
        chans, ds = [ch], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                cout = int(mult * model_channels)
                self.input_blocks.append(Stage(*level_layers(ch, cout, ds)))
                ch = cout
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(Stage(res(ch, ch, down=True)))
                chans.append(ch)
                ds *= 2
        mid = [res(ch, ch)]
        if temporal_block:
            mid.append(Wrapped(res(ch, ch, 3)))
        mid.append(attn(ch, bottleneck=True))
        if temporal_block:
            mid.append(Wrapped(TemporalAttention(ch, 5, num_heads, num_head_channels)))
        mid.append(res(ch, ch))
        if temporal_block:
            mid.append(Wrapped(res(ch, ch, 3)))
        self.middle_block = Stage(*mid)
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                cout = int(model_channels * mult)
                layers = level_layers(ch + chans.pop(), cout, ds)
                ch = cout
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True))
                    ds //= 2
                self.output_blocks.append(Stage(*layers))
        self.out = nn.Sequential(Wrapped(nn.GroupNorm(32, ch)), nn.SiLU(),
                                 _zero(Wrapped(nn.Conv2d(int(channel_mult[0] * model_channels),
                                                         out_channels, 3, padding=1))))

    @torch.no_grad()
    def compute_flow(self, lqs):
        """unet_new.py:1283-1309."""
        lqs = ((lqs + 1) / 2).clamp(0, 1)
        n, t, c, h, w = lqs.shape
        a = lqs[:, :-1].reshape(-1, c, h, w)
        b = lqs[:, 1:].reshape(-1, c, h, w)
        flows_backward = self.spynet(a, b).view(n, t - 1, 2, h, w)
        flows_forward = self.spynet(b, a).view(n, t - 1, 2, h, w)
        return flows_forward, flows_backward

    def forward(self, x, timesteps, low_res_input=None, num_frames=None, rnn_input=None,
                enable_cross_frames=True, vsrpp_weights=None, **kwargs):
        """unet_new.py:1311-1362."""
        x = x.reshape(-1, num_frames, *x.shape[1:])
        x = torch.cat([x, low_res_input], dim=2)
        if rnn_input is None:
            rnn_input = low_res_input
        flows = {}
        for r in self.need_flows_res:
            if rnn_input.shape[-1] != r:
                fi = per_frame(rnn_input, lambda z: F.interpolate(z, (r, r), mode="bicubic"))
            else:
                fi = rnn_input
            flows[r] = self.compute_flow(fi)
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels))
        h, hs = x, []
        for blk in self.input_blocks:
            h = blk(h, emb, flows, vsrpp_weights, enable_cross_frames)
            hs.append(h)
        h = self.middle_block(h, emb, flows, vsrpp_weights, enable_cross_frames)
        for blk in self.output_blocks:
            h = blk(torch.cat([h, hs.pop()], dim=2), emb, flows, vsrpp_weights, enable_cross_frames)
        h = F.silu(group_norm_over_clip(h, self.out[0].wrapped_module))
        h = per_frame(h, self.out[2].wrapped_module)
        return h.reshape(-1, *h.shape[2:])
