"""ParseNet face parsing on the HIP kernels (parsing half).

Mirror of the reference's ``guided_diffusion/facelib/parsing/parsenet.py`` (``ParseNet(in_size, out_size,
min_feat_size, base_ch, parsing_ch, res_depth, relu_type, norm_type, ch_range)``, ``forward(x) ->
(out_mask, out_img)``), the network ``FaceRestoreHelper.face_parse`` holds (face_restoration_helper.py:118;
built by facelib/parsing/__init__.py:13-14 as ``ParseNet(in_size=512, out_size=512, parsing_ch=19)``).  On the
sampling path its class-0 mask becomes the per-pixel ``vsrpp_weights`` of the bicubic tasks
(scripts/video_sample.py:427-444): ``face_weight`` below is that expression, evaluated on the GPU.

The modules are parameter containers with the reference's state-dict names (``parsing_parsenet.pth`` loads
unchanged).  Every ConvLayer -- [nearest x2] -> ReflectionPad2d(1) -> Conv2d(3x3, stride 1 | 2) -> [BatchNorm2d,
eval] -> [LeakyReLU(0.2)] -- is ONE ``flair_conv_nhwc`` launch: reflection padding is a conv parameter
(``reflect_pad``), the eval-mode BatchNorm is folded into the packed weights and bias, the activation and the
residual adds sit in the conv epilogue.  It runs once per window (not per denoising step), in float32.

RetinaFace detection / landmark alignment (the other half of row 4) is not provided.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .. import ops as A


class NormLayer(nn.Module):
    """parsenet.py:8-41 ('bn' and 'none' are the types ParseNet instantiates)."""

    def __init__(self, channels, normalize_shape=None, norm_type="bn"):
        super().__init__()
        self.norm_type = norm_type.lower()
        if self.norm_type == "bn":
            self.norm = nn.BatchNorm2d(channels, affine=True)
        elif self.norm_type != "none":
            raise NotImplementedError(f"flair_amd: ParseNet norm type {norm_type!r}")


class ConvLayer(nn.Module):
    """parsenet.py:74-105."""

    def __init__(self, in_channels, out_channels, kernel_size=3, scale="none", norm_type="none", relu_type="none",
                 use_pad=True, bias=True):
        super().__init__()
        if kernel_size != 3 or not use_pad:
            raise NotImplementedError("flair_amd: ParseNet uses reflect-padded 3x3 convolutions only")
        relu_type = relu_type.lower()
        if relu_type not in ("leakyrelu", "none"):
            raise NotImplementedError(f"flair_amd: ParseNet relu type {relu_type!r}")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.scale, self.act = scale, relu_type == "leakyrelu"
        if norm_type in ("bn",):
            bias = False
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, 2 if scale == "down" else 1, bias=bias)
        self.norm = NormLayer(out_channels, norm_type=norm_type)

    def pack(self, dtype, device):
        w = self.conv2d.weight.detach().float()
        b = self.conv2d.bias.detach().float() if self.conv2d.bias is not None else w.new_zeros(self.out_channels)
        if self.norm.norm_type == "bn":                       # eval-mode BatchNorm folded into the convolution
            bn = self.norm.norm
            g = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            w = w * g.view(-1, 1, 1, 1)
            b = (b - bn.running_mean.detach().float()) * g + bn.bias.detach().float()
        cpad = (self.out_channels + 3) // 4 * 4
        cin = self.in_channels
        self._w = ops.pack_conv_weight(w.to(device), [(cin, ops.pad_channels(cin, dtype))], dtype, cpad)
        b = b.to(device)
        self._b = torch.cat([b, b.new_zeros(cpad - self.out_channels)]).contiguous()
        self._cout = cpad

    def run(self, x, res0=None, res1=None):
        if self.scale == "up":
            x = ops.resize(x, (2 * x.shape[1], 2 * x.shape[2]), 4)
        return ops.conv(x, self._w, self._b, self._cout, (1, 3, 3), stride=2 if self.scale == "down" else 1,
                        reflect_pad=True, act=A.ACT_LRELU02 if self.act else A.ACT_NONE, res0=res0, res1=res1)


class ResidualBlock(nn.Module):
    """parsenet.py:108-128."""

    def __init__(self, c_in, c_out, relu_type="prelu", norm_type="bn", scale="none"):
        super().__init__()
        if not (scale == "none" and c_in == c_out):
            self.shortcut_func = ConvLayer(c_in, c_out, 3, scale)
        s1, s2 = {"down": ("none", "down"), "up": ("up", "none"), "none": ("none", "none")}[scale]
        self.conv1 = ConvLayer(c_in, c_out, 3, s1, norm_type=norm_type, relu_type=relu_type)
        self.conv2 = ConvLayer(c_out, c_out, 3, s2, norm_type=norm_type, relu_type="none")

    def run(self, x, extra=None):
        ident = self.shortcut_func.run(x) if hasattr(self, "shortcut_func") else x
        return self.conv2.run(self.conv1.run(x), res0=ident, res1=extra)


class ParseNet(nn.Module):
    """parsenet.py:131-194."""

    def __init__(self, in_size=128, out_size=128, min_feat_size=32, base_ch=64, parsing_ch=19, res_depth=10,
                 relu_type="LeakyReLU", norm_type="bn", ch_range=(32, 256)):
        super().__init__()
        self.res_depth, self.parsing_ch = res_depth, parsing_ch
        act_args = {"norm_type": norm_type, "relu_type": relu_type}
        min_ch, max_ch = ch_range

        def ch_clip(c):
            return max(min_ch, min(c, max_ch))
        min_feat_size = min(in_size, min_feat_size)
        down_steps = int(math.log2(in_size // min_feat_size))
        up_steps = int(math.log2(out_size // min_feat_size))
        encoder = [ConvLayer(3, base_ch, 3, 1)]                 # the reference passes scale=1: no rescaling
        head_ch = base_ch
        for _ in range(down_steps):
            encoder.append(ResidualBlock(ch_clip(head_ch), ch_clip(head_ch * 2), scale="down", **act_args))
            head_ch *= 2
        body = [ResidualBlock(ch_clip(head_ch), ch_clip(head_ch), **act_args) for _ in range(res_depth)]
        decoder = []
        for _ in range(up_steps):
            decoder.append(ResidualBlock(ch_clip(head_ch), ch_clip(head_ch // 2), scale="up", **act_args))
            head_ch //= 2
        self.encoder = nn.Sequential(*encoder)
        self.body = nn.Sequential(*body)
        self.decoder = nn.Sequential(*decoder)
        self.out_img_conv = ConvLayer(ch_clip(head_ch), 3)
        self.out_mask_conv = ConvLayer(ch_clip(head_ch), parsing_ch)
        self.dtype = torch.float32
        self._packed_key = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._packed_key = None                 # folded / packed weights are rebuilt on the next forward
        return out

    def _ensure_packed(self, device):
        key = (self.dtype, device)
        if self._packed_key != key:
            for m in self.modules():
                if isinstance(m, ConvLayer):
                    m.pack(self.dtype, device)
            self._packed_key = key

    def _features(self, x):
        """(B, 3, H, W) f32 in [-1, 1] -> decoder output clip tensor (B, H, W, C)."""
        self._ensure_packed(x.device)
        B, _, H, W = x.shape
        h = torch.zeros((B, H, W, ops.pad_channels(3, self.dtype)), dtype=self.dtype, device=x.device)
        ops.nchw_to_clip(x.float().contiguous(), h, 0)
        feat = self.encoder[0].run(h)
        for blk in list(self.encoder)[1:]:
            feat = blk.run(feat)
        y = feat
        for i, blk in enumerate(self.body):                     # x = feat + body(feat): the add rides on the last block
            y = blk.run(y, extra=feat if i == len(self.body) - 1 else None)
        if len(self.body) == 0:
            y = ops.axpby(feat, feat, 1.0, 1.0)
        for blk in self.decoder:
            y = blk.run(y)
        return y

    @torch.no_grad()
    def forward(self, x):
        y = self._features(x)
        mask = self.out_mask_conv.run(y)
        img = self.out_img_conv.run(y)
        return ops.clip_to_nchw(mask, self.parsing_ch), ops.clip_to_nchw(img, 3)

    @torch.no_grad()
    def face_weight(self, frames, w_face):
        """``mask * w_face + (1 - mask)`` with ``mask = (face_parse(frames)[0].argmax(1, keepdim=True) == 0)``
        (scripts/video_sample.py:427-444; w_face 0.93 for x8, 0.98 for x16 bicubic): (T, 1, H, W) float32, computed
        without leaving the GPU (the arg-max kernel looks the weight up in a 19-entry table)."""
        logits = self.out_mask_conv.run(self._features(frames))
        table = torch.ones((self.parsing_ch, 1), dtype=torch.float32, device=frames.device)
        table[0, 0] = float(w_face)
        w, _ = ops.argmax_codebook(logits, self.parsing_ch, table)
        return w.permute(0, 3, 1, 2).contiguous()
