"""CodeFormer auxiliary prior on the HIP kernels.

Mirror of the reference's ``guided_diffusion/codeformer.py``: ``CodeFormer(dim_embd, n_head, n_layers,
codebook_size, latent_size, connect_list)`` (:600-681) with ``forward(x, w=0, detach_16=True,
code_only=False, adain=False) -> (out, logits, lq_feat)`` (:692-753), called by the sampler as
``aux_model(pred_xstart, t, x)`` = ``gan(x0, w=1.0, adain=True)[0]`` (scripts/video_sample.py:450-452,
gaussian_diffusion.py:471-496).  The modules below are parameter containers that reproduce the
reference's state-dict names (``encoder.blocks.N.conv1.weight`` ...), so ``codeformer.pth``
(``params_ema``) loads unchanged through ``checkpoint.load_reference_checkpoint``; the arithmetic runs
on libflair_hip.so through ``flair_amd.ops`` on NHWC tensors:

  * 3x3 / 1x1 convolutions, the asymmetric stride-2 ``Downsample`` and all Linear layers (1x1 on the
    16x16 token grid): ``flair_conv_nhwc`` (MFMA);  GroupNorm(32, eps 1e-6)(+swish): ``flair_groupnorm_nhwc``;
  * transformer: ``flair_layernorm_nhwc`` (+ positional add), 8 x 64 heads on ``flair_qkv_attention``,
    GELU in the conv epilogue;  AttnBlock (one head of 512): ``flair_attention_wide``;
  * code arg-max + codebook lookup, AdaIN, SFT fusion: ``flair_argmax_codebook`` / ``flair_adain_nhwc`` /
    ``flair_sft_fuse``.

The reference runs the prior in fp32 (only the UNet is converted to fp16): ``dtype`` is float32 by default;
``convert_to_bf16()`` switches the activations / conv weights to bf16 (statistics stay f32).
Frames are a batch: no cross-frame arithmetic exists in the prior.
"""
import torch
import torch.nn as nn

from .. import ops
from .. import ops as A

CH_MULT = (1, 2, 2, 4, 4, 8)
GN_EPS = 1e-6


def _dev(p, device):
    return p.detach().to(device=device, dtype=torch.float32).contiguous()


def _pack(w, dtype, device, segs=None, cout_pad=None):
    """(Cout, Cin, kh, kw) or (Cout, Cin) f32 -> packed [Cout][taps][Cin] in ``dtype``."""
    w = w.detach().to(device)
    if w.dim() == 2:
        w = w[:, :, None, None]
    cin = w.shape[1]
    return ops.pack_conv_weight(w, segs or [(cin, ops.pad_channels(cin, dtype))], dtype, cout_pad)


def normalize(in_channels):
    """codeformer.py:9-12."""
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=GN_EPS, affine=True)


def _gn(x, pk, name, act=A.ACT_NONE, x1=None):
    return ops.group_norm(x, pk[name + "_g"], pk[name + "_b"], x1=x1, eps=GN_EPS, act=act, frames_per_stat=1)


class VectorQuantizer(nn.Module):
    """codeformer.py:21-94 -- only the codebook is used at inference (get_codebook_feat, via the transformer's
    indices); the nearest-neighbour ``forward`` belongs to VQGAN training."""

    def __init__(self, codebook_size, emb_dim, beta):
        super().__init__()
        self.codebook_size, self.emb_dim, self.beta = codebook_size, emb_dim, beta
        self.embedding = nn.Embedding(codebook_size, emb_dim)
        self.embedding.weight.data.uniform_(-1.0 / codebook_size, 1.0 / codebook_size)


class Downsample(nn.Module):
    """codeformer.py:138-149: F.pad(x, (0, 1, 0, 1)) + 3x3 stride-2 conv without padding."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def pack(self, dtype, device):
        self._pk = dict(w=_pack(self.conv.weight, dtype, device), b=_dev(self.conv.bias, device))

    def run(self, x):
        return ops.conv(x, self._pk["w"], self._pk["b"], self.conv.out_channels, (1, 3, 3), stride=2, asym_pad=True)


class Upsample(nn.Module):
    """codeformer.py:152-163: nearest x2, then 3x3 conv."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def pack(self, dtype, device):
        self._pk = dict(w=_pack(self.conv.weight, dtype, device), b=_dev(self.conv.bias, device))

    def run(self, x):
        up = ops.resize(x, (2 * x.shape[1], 2 * x.shape[2]), 4)
        return ops.conv(up, self._pk["w"], self._pk["b"], self.conv.out_channels, (1, 3, 3))


class ResBlock(nn.Module):
    """codeformer.py:166-195."""

    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, self.out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = normalize(self.out_channels)
        self.conv2 = nn.Conv2d(self.out_channels, self.out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            self.conv_out = nn.Conv2d(in_channels, self.out_channels, kernel_size=1, stride=1, padding=0)

    def pack(self, dtype, device, split=None):
        """split: widths of the two implicitly concatenated input segments (Fuse_sft_block.encode_enc)."""
        segs = [(s, s) for s in split] if split else None
        self._pk = dict(w1=_pack(self.conv1.weight, dtype, device), b1=_dev(self.conv1.bias, device),
                        w2=_pack(self.conv2.weight, dtype, device), b2=_dev(self.conv2.bias, device),
                        n1_g=_dev(self.norm1.weight, device), n1_b=_dev(self.norm1.bias, device),
                        n2_g=_dev(self.norm2.weight, device), n2_b=_dev(self.norm2.bias, device))
        if self.in_channels != self.out_channels:
            self._pk["ws"] = _pack(self.conv_out.weight, dtype, device, segs)
            self._pk["bs"] = _dev(self.conv_out.bias, device)

    def run(self, x, x1=None):
        pk, co = self._pk, self.out_channels
        h = _gn(x, pk, "n1", A.ACT_SILU, x1=x1)
        h = ops.conv(h, pk["w1"], pk["b1"], co, (1, 3, 3))
        h = _gn(h, pk, "n2", A.ACT_SILU)
        if "ws" in pk:
            skip = ops.conv([x] if x1 is None else [x, x1], pk["ws"], pk["bs"], co, (1, 1, 1))
        else:
            assert x1 is None
            skip = x
        return ops.conv(h, pk["w2"], pk["b2"], co, (1, 3, 3), res0=skip)


class AttnBlock(nn.Module):
    """codeformer.py:198-241: GroupNorm, 1x1 q / k / v, one attention head of width C, 1x1 proj_out, residual."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def pack(self, dtype, device):
        wqkv = torch.cat([self.q.weight, self.k.weight, self.v.weight], dim=0)
        self._pk = dict(n_g=_dev(self.norm.weight, device), n_b=_dev(self.norm.bias, device),
                        wqkv=_pack(wqkv, dtype, device),
                        bqkv=_dev(torch.cat([self.q.bias, self.k.bias, self.v.bias]), device),
                        wp=_pack(self.proj_out.weight, dtype, device), bp=_dev(self.proj_out.bias, device))

    def run(self, x):
        pk, c = self._pk, self.in_channels
        qkv = ops.conv(_gn(x, pk, "n"), pk["wqkv"], pk["bqkv"], 3 * c, (1, 1, 1))
        a = ops.attention_wide(qkv, 1, c, q_off=0, k_off=c, v_off=2 * c, head_stride=c)
        return ops.conv(a, pk["wp"], pk["bp"], c, (1, 1, 1), res0=x)


class _Conv(nn.Conv2d):
    """A bare 3x3 convolution entry of encoder.blocks / generator.blocks (conv_in / conv_out)."""

    def pack(self, dtype, device):
        cpad = (self.out_channels + 3) // 4 * 4
        b = _dev(self.bias, device)
        if cpad != self.out_channels:
            b = torch.cat([b, b.new_zeros(cpad - self.out_channels)]).contiguous()
        self._pk = dict(w=_pack(self.weight, dtype, device, cout_pad=cpad), b=b, cout=cpad)

    def run(self, x):
        return ops.conv(x, self._pk["w"], self._pk["b"], self._pk["cout"], (1, 3, 3))


class _Norm(nn.GroupNorm):
    """The final ``normalize`` entry (no activation follows it: codeformer.py:288-292, :341-347)."""

    def pack(self, dtype, device):
        self._pk = dict(n_g=_dev(self.weight, device), n_b=_dev(self.bias, device))

    def run(self, x):
        return _gn(x, self._pk, "n")


class Encoder(nn.Module):
    """codeformer.py:244-299."""

    def __init__(self, in_channels, nf, emb_dim, ch_mult, num_res_blocks, resolution, attn_resolutions):
        super().__init__()
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        blocks = [_Conv(in_channels, nf, kernel_size=3, stride=1, padding=1)]
        block_in_ch = nf
        for i in range(len(ch_mult)):
            block_in_ch = nf * in_ch_mult[i]
            block_out_ch = nf * ch_mult[i]
            for _ in range(num_res_blocks):
                blocks.append(ResBlock(block_in_ch, block_out_ch))
                block_in_ch = block_out_ch
                if curr_res in attn_resolutions:
                    blocks.append(AttnBlock(block_in_ch))
            if i != len(ch_mult) - 1:
                blocks.append(Downsample(block_in_ch))
                curr_res //= 2
        blocks += [ResBlock(block_in_ch, block_in_ch), AttnBlock(block_in_ch), ResBlock(block_in_ch, block_in_ch)]
        n = _Norm(32, block_in_ch, eps=GN_EPS, affine=True)
        blocks += [n, _Conv(block_in_ch, emb_dim, kernel_size=3, stride=1, padding=1)]
        self.blocks = nn.ModuleList(blocks)


class Generator(nn.Module):
    """codeformer.py:302-354."""

    def __init__(self, nf, emb_dim, ch_mult, res_blocks, img_size, attn_resolutions):
        super().__init__()
        block_in_ch = nf * ch_mult[-1]
        curr_res = img_size // 2 ** (len(ch_mult) - 1)
        blocks = [_Conv(emb_dim, block_in_ch, kernel_size=3, stride=1, padding=1),
                  ResBlock(block_in_ch, block_in_ch), AttnBlock(block_in_ch), ResBlock(block_in_ch, block_in_ch)]
        for i in reversed(range(len(ch_mult))):
            block_out_ch = nf * ch_mult[i]
            for _ in range(res_blocks):
                blocks.append(ResBlock(block_in_ch, block_out_ch))
                block_in_ch = block_out_ch
                if curr_res in attn_resolutions:
                    blocks.append(AttnBlock(block_in_ch))
            if i != 0:
                blocks.append(Upsample(block_in_ch))
                curr_res *= 2
        blocks += [_Norm(32, block_in_ch, eps=GN_EPS, affine=True),
                   _Conv(block_in_ch, 3, kernel_size=3, stride=1, padding=1)]
        self.blocks = nn.ModuleList(blocks)


class TransformerSALayer(nn.Module):
    """codeformer.py:531-571: pre-norm self-attention (q = k = norm(x) + pos, v = norm(x)) and GELU MLP."""

    def __init__(self, embed_dim, nhead=8, dim_mlp=2048, dropout=0.0, activation="gelu"):
        super().__init__()
        if activation != "gelu" or dropout != 0.0:
            raise NotImplementedError("flair_amd: CodeFormer's transformer uses GELU and dropout 0")
        self.embed_dim, self.nhead, self.dim_mlp = embed_dim, nhead, dim_mlp
        self.self_attn = nn.MultiheadAttention(embed_dim, nhead, dropout=dropout)
        self.linear1 = nn.Linear(embed_dim, dim_mlp)
        self.linear2 = nn.Linear(dim_mlp, embed_dim)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)

    def pack(self, dtype, device):
        e, at = self.embed_dim, self.self_attn
        self._pk = dict(
            wqk=_pack(at.in_proj_weight[:2 * e], dtype, device), bqk=_dev(at.in_proj_bias[:2 * e], device),
            wv=_pack(at.in_proj_weight[2 * e:], dtype, device), bv=_dev(at.in_proj_bias[2 * e:], device),
            wo=_pack(at.out_proj.weight, dtype, device), bo=_dev(at.out_proj.bias, device),
            w1=_pack(self.linear1.weight, dtype, device), b1=_dev(self.linear1.bias, device),
            w2=_pack(self.linear2.weight, dtype, device), b2=_dev(self.linear2.bias, device),
            n1_g=_dev(self.norm1.weight, device), n1_b=_dev(self.norm1.bias, device),
            n2_g=_dev(self.norm2.weight, device), n2_b=_dev(self.norm2.bias, device))

    def run(self, tgt, pos):
        pk, e = self._pk, self.embed_dim
        t2, t2pos = ops.layer_norm(tgt, pk["n1_g"], pk["n1_b"], eps=self.norm1.eps, pos=pos)
        qkv = torch.empty(tgt.shape[:3] + (3 * e,), dtype=tgt.dtype, device=tgt.device)
        ops.conv(t2pos, pk["wqk"], pk["bqk"], 2 * e, (1, 1, 1), out=qkv[..., :2 * e])
        ops.conv(t2, pk["wv"], pk["bv"], e, (1, 1, 1), out=qkv[..., 2 * e:])
        a = ops.qkv_attention(qkv, self.nhead, new_order=True)
        tgt = ops.conv(a, pk["wo"], pk["bo"], e, (1, 1, 1), res0=tgt)
        t2 = ops.layer_norm(tgt, pk["n2_g"], pk["n2_b"], eps=self.norm2.eps)
        h = ops.conv(t2, pk["w1"], pk["b1"], self.dim_mlp, (1, 1, 1), act=A.ACT_GELU)
        return ops.conv(h, pk["w2"], pk["b2"], e, (1, 1, 1), res0=tgt)


class Fuse_sft_block(nn.Module):
    """codeformer.py:574-597: out = dec + w * (dec * scale(enc) + shift(enc)), enc = ResBlock(cat[enc, dec])."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.in_ch, self.out_ch = in_ch, out_ch
        self.encode_enc = ResBlock(2 * in_ch, out_ch)
        self.scale = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size=3, padding=1), nn.LeakyReLU(0.2, True),
                                   nn.Conv2d(out_ch, out_ch, kernel_size=3, padding=1))
        self.shift = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size=3, padding=1), nn.LeakyReLU(0.2, True),
                                   nn.Conv2d(out_ch, out_ch, kernel_size=3, padding=1))

    def pack(self, dtype, device):
        self.encode_enc.pack(dtype, device, split=[self.in_ch, self.in_ch])
        # the first convolutions of both branches read the same tensor: one launch, 2 * out_ch outputs
        self._pk = dict(
            w0=_pack(torch.cat([self.scale[0].weight, self.shift[0].weight], dim=0), dtype, device),
            b0=_dev(torch.cat([self.scale[0].bias, self.shift[0].bias]), device),
            ws=_pack(self.scale[2].weight, dtype, device), bs=_dev(self.scale[2].bias, device),
            wh=_pack(self.shift[2].weight, dtype, device), bh=_dev(self.shift[2].bias, device))

    def run(self, enc_feat, dec_feat, w=1.0):
        pk, co = self._pk, self.out_ch
        enc = self.encode_enc.run(enc_feat, x1=dec_feat)
        mid = ops.conv(enc, pk["w0"], pk["b0"], 2 * co, (1, 3, 3), act=A.ACT_LRELU02)
        scale = ops.conv(mid[..., :co], pk["ws"], pk["bs"], co, (1, 3, 3))
        shift = ops.conv(mid[..., co:], pk["wh"], pk["bh"], co, (1, 3, 3))
        return ops.sft_fuse(dec_feat, scale, shift, float(w))


class VQAutoEncoder(nn.Module):
    """codeformer.py:357-434 ('nearest' quantiser; the Gumbel variant and ``model_path`` loading are training-side)."""

    def __init__(self, img_size, nf, ch_mult, quantizer="nearest", res_blocks=2, attn_resolutions=(16,),
                 codebook_size=1024, emb_dim=256, beta=0.25, gumbel_straight_through=False, gumbel_kl_weight=1e-8,
                 model_path=None):
        super().__init__()
        if quantizer != "nearest":
            raise NotImplementedError("flair_amd: CodeFormer uses the 'nearest' quantiser")
        self.in_channels, self.nf, self.n_blocks = 3, nf, res_blocks
        self.codebook_size, self.embed_dim, self.ch_mult = codebook_size, emb_dim, ch_mult
        self.resolution, self.attn_resolutions = img_size, list(attn_resolutions)
        self.encoder = Encoder(3, nf, emb_dim, ch_mult, res_blocks, img_size, self.attn_resolutions)
        self.quantize = VectorQuantizer(codebook_size, emb_dim, beta)
        self.generator = Generator(nf, emb_dim, ch_mult, res_blocks, img_size, self.attn_resolutions)
        if model_path is not None:                             # {'params_ema' | 'params': state_dict}, :415-428
            from ..checkpoint import load_reference_checkpoint
            load_reference_checkpoint(self, model_path)


class CodeFormer(VQAutoEncoder):
    """codeformer.py:600-753."""

    def __init__(self, dim_embd=512, n_head=8, n_layers=9, codebook_size=1024, latent_size=256,
                 connect_list=("32", "64", "128", "256"), fix_modules=("quantize", "generator"), vqgan_path=None):
        super().__init__(512, 64, list(CH_MULT), "nearest", 2, [16], codebook_size)
        if vqgan_path is not None:
            from ..checkpoint import load_reference_checkpoint
            load_reference_checkpoint(self, vqgan_path, strict=False)
        for module in fix_modules or ():
            for p in getattr(self, module).parameters():
                p.requires_grad = False
        self.connect_list = list(connect_list)
        self.n_layers, self.dim_embd, self.dim_mlp = n_layers, dim_embd, dim_embd * 2
        self.position_emb = nn.Parameter(torch.zeros(latent_size, dim_embd))
        self.feat_emb = nn.Linear(256, dim_embd)
        self.ft_layers = nn.Sequential(*[TransformerSALayer(embed_dim=dim_embd, nhead=n_head, dim_mlp=self.dim_mlp,
                                                            dropout=0.0) for _ in range(n_layers)])
        self.idx_pred_layer = nn.Sequential(nn.LayerNorm(dim_embd), nn.Linear(dim_embd, codebook_size, bias=False))
        self.channels = {"16": 512, "32": 256, "64": 256, "128": 128, "256": 128, "512": 64}
        self.fuse_encoder_block = {"512": 2, "256": 5, "128": 8, "64": 11, "32": 14, "16": 18}
        self.fuse_generator_block = {"16": 6, "32": 9, "64": 12, "128": 15, "256": 18, "512": 21}
        self.fuse_convs_dict = nn.ModuleDict()
        for f_size in self.connect_list:
            in_ch = self.channels[f_size]
            self.fuse_convs_dict[f_size] = Fuse_sft_block(in_ch, in_ch)
        self.dtype = torch.float32
        self._packed_key = None

    def convert_to_bf16(self):
        self.dtype = torch.bfloat16
        self._packed_key = None
        return self

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._packed_key = None                 # kernel-native weight copies are rebuilt on the next forward
        return out

    def _ensure_packed(self, device):
        key = (self.dtype, device)
        if self._packed_key == key:
            return
        dt = self.dtype
        fused_inputs = {id(f.encode_enc) for f in self.fuse_convs_dict.values()}   # packed by their Fuse_sft_block
        for m in self.modules():
            if hasattr(m, "pack") and m is not self and id(m) not in fused_inputs:
                m.pack(dt, device)
        self._pk = dict(
            pos=_dev(self.position_emb, device),
            wf=_pack(self.feat_emb.weight, dt, device), bf=_dev(self.feat_emb.bias, device),
            hn_g=_dev(self.idx_pred_layer[0].weight, device), hn_b=_dev(self.idx_pred_layer[0].bias, device),
            wi=_pack(self.idx_pred_layer[1].weight, dt, device),
            codebook=_dev(self.quantize.embedding.weight, device))
        self._packed_key = key

    @torch.no_grad()
    def forward(self, x, w=0, detach_16=True, code_only=False, adain=False, code_idx=None):
        """x: (B, 3, 512, 512) in [-1, 1] on the GPU -> (out (B, 3, 512, 512), logits (B, 256, codebook),
        lq_feat (B, 256, 16, 16)), all float32 NCHW like the reference.  ``code_idx`` ((B, 256) integer tensor)
        replaces the arg-max of the logits (tests)."""
        if x.shape[1:] != (3, 512, 512):
            raise ValueError(f"CodeFormer works on aligned 512x512 faces (codeformer.py:730), got {tuple(x.shape)}")
        dev, dt = x.device, self.dtype
        self._ensure_packed(dev)
        pk = self._pk
        B = x.shape[0]
        h = torch.zeros((B, 512, 512, ops.pad_channels(3, dt)), dtype=dt, device=dev)
        ops.nchw_to_clip(x.float().contiguous(), h, 0)
        # ---- encoder (:694-701)
        enc_feat = {}
        taps = {self.fuse_encoder_block[s] for s in self.connect_list}
        for i, blk in enumerate(self.encoder.blocks):
            h = blk.run(h)
            if i in taps:
                enc_feat[str(h.shape[2])] = h
        lq = h                                                     # (B, 16, 16, 256)
        # ---- transformer over the 256 tokens (:703-714); Linear layers are 1x1 convolutions on the token grid
        q = ops.conv(lq, pk["wf"], pk["bf"], self.dim_embd, (1, 1, 1))
        for layer in self.ft_layers:
            q = layer.run(q, pk["pos"])
        t = ops.layer_norm(q, pk["hn_g"], pk["hn_b"], eps=self.idx_pred_layer[0].eps)
        logits = ops.conv(t, pk["wi"], None, self.codebook_size, (1, 1, 1))
        logits_out = logits.float().reshape(B, 256, self.codebook_size)
        lq_out = ops.clip_to_nchw(lq, 256)
        if code_only:
            return logits_out, lq_out
        # ---- code lookup + AdaIN (:727-741)
        forced = None if code_idx is None else code_idx.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
        quant, _ = ops.argmax_codebook(logits, self.codebook_size, pk["codebook"], forced_idx=forced)
        if adain:
            quant = ops.adain(quant, lq)
        # ---- generator with SFT fusion (:743-751)
        h = quant
        fuse = {self.fuse_generator_block[s] for s in self.connect_list}
        for i, blk in enumerate(self.generator.blocks):
            h = blk.run(h)
            if i in fuse and w > 0:
                s = str(h.shape[2])
                h = self.fuse_convs_dict[s].run(enc_feat[s], h, w)
        return ops.clip_to_nchw(h, 3), logits_out, lq_out
