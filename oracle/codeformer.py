"""CPU restatement of the CodeFormer auxiliary prior in plain PyTorch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows guided_diffusion/codeformer.py of the
reference: ``CodeFormer.forward`` (:692-753) with the configuration scripts/video_sample.py:351-357 builds
(VQAutoEncoder(512, 64, [1, 2, 2, 4, 4, 8], 'nearest', 2, [16], 1024), 9 transformer layers of width 512,
8 heads, SFT fusion at 32 / 64 / 128 / 256).  It is written as functions over a STATE DICT with the
reference's parameter names, so the same name-seeded weights drive the reference (fixture generation),
this oracle and the HIP module.  Pinned by tests/golden/g9_codeformer.npz (tests/test_codeformer.py).

Only the inference path the sampler uses is restated: ``code_only=False``, ``detach_16`` irrelevant
without autograd, quantiser 'nearest' (its own forward is never called: the code indices come from the
transformer's logits, :727-732).
"""
import torch
import torch.nn.functional as F

CH_MULT = (1, 2, 2, 4, 4, 8)                      # codeformer.py:612-614
NF = 64
N_LAYERS, N_HEAD, DIM = 9, 8, 512                 # scripts/video_sample.py:351-357
CONNECT = ("32", "64", "128", "256")
FUSE_ENCODER_BLOCK = {"512": 2, "256": 5, "128": 8, "64": 11, "32": 14, "16": 18}      # codeformer.py:654-661
FUSE_GENERATOR_BLOCK = {"16": 6, "32": 9, "64": 12, "128": 15, "256": 18, "512": 21}   # codeformer.py:663-670


def _gn(sd, name, x):
    """normalize(): GroupNorm(32, C, eps=1e-6) -- codeformer.py:9-12."""
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], 1e-6)


def _conv(sd, name, x, padding=1, stride=1):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _swish(x):
    return x * torch.sigmoid(x)                    # codeformer.py:15-17


def res_block(sd, name, x_in):
    """ResBlock.forward -- codeformer.py:184-195 (1x1 ``conv_out`` skip only when the width changes)."""
    x = _conv(sd, name + ".conv1", _swish(_gn(sd, name + ".norm1", x_in)))
    x = _conv(sd, name + ".conv2", _swish(_gn(sd, name + ".norm2", x)))
    if name + ".conv_out.weight" in sd:
        x_in = _conv(sd, name + ".conv_out", x_in, padding=0)
    return x + x_in


def attn_block(sd, name, x):
    """AttnBlock.forward -- codeformer.py:217-241: single-head attention over the h*w pixels, width C."""
    h = _gn(sd, name + ".norm", x)
    q, k, v = (_conv(sd, f"{name}.{n}", h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    w_ = torch.bmm(q.reshape(b, c, -1).permute(0, 2, 1), k.reshape(b, c, -1)) * (int(c) ** -0.5)
    w_ = F.softmax(w_, dim=2)
    out = torch.bmm(v.reshape(b, c, -1), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, name + ".proj_out", out, padding=0)


def downsample(sd, name, x):
    """Downsample.forward -- codeformer.py:145-149: zero row / column appended, 3x3 stride-2 conv, no padding."""
    return _conv(sd, name + ".conv", F.pad(x, (0, 1, 0, 1)), padding=0, stride=2)


def upsample(sd, name, x):
    """Upsample.forward -- codeformer.py:159-163."""
    return _conv(sd, name + ".conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))


def block_kinds(prefix, sd):
    """Kind of each entry of ``encoder.blocks`` / ``generator.blocks`` from the parameter names alone
    (the constructors at codeformer.py:262-293 / :316-348 interleave convs, ResBlocks, AttnBlocks,
    Down/Upsample and the final norm)."""
    kinds, i = [], 0
    while any(k.startswith(f"{prefix}.{i}.") for k in sd):
        p = f"{prefix}.{i}"
        if p + ".norm1.weight" in sd:
            kinds.append("res")
        elif p + ".proj_out.weight" in sd:
            kinds.append("attn")
        elif p + ".conv.weight" in sd:
            kinds.append("resample")
        elif sd[p + ".weight"].dim() == 1:
            kinds.append("norm")
        else:
            kinds.append("conv")
        i += 1
    return kinds


def run_block(sd, prefix, i, kind, x, down):
    p = f"{prefix}.{i}"
    if kind == "res":
        return res_block(sd, p, x)
    if kind == "attn":
        return attn_block(sd, p, x)
    if kind == "resample":
        return downsample(sd, p, x) if down else upsample(sd, p, x)
    if kind == "norm":
        return _gn(sd, p, x)
    return _conv(sd, p, x)


def transformer_layer(sd, name, tgt, pos):
    """TransformerSALayer.forward -- codeformer.py:552-571 (pre-norm; q = k = norm(x) + pos, v = norm(x);
    nn.MultiheadAttention with 8 heads of 64; GELU MLP 512 -> 1024 -> 512; dropout 0)."""
    L, B, E = tgt.shape
    t2 = F.layer_norm(tgt, (E,), sd[name + ".norm1.weight"], sd[name + ".norm1.bias"])
    wi, bi = sd[name + ".self_attn.in_proj_weight"], sd[name + ".self_attn.in_proj_bias"]
    qk_in = t2 + pos
    q = F.linear(qk_in, wi[:E], bi[:E])
    k = F.linear(qk_in, wi[E:2 * E], bi[E:2 * E])
    v = F.linear(t2, wi[2 * E:], bi[2 * E:])
    d = E // N_HEAD

    def heads(t):                                   # (L, B, E) -> (B * heads, L, d)
        return t.reshape(L, B * N_HEAD, d).transpose(0, 1)

    a = torch.softmax(torch.bmm(heads(q) * d ** -0.5, heads(k).transpose(1, 2)), dim=-1)
    o = torch.bmm(a, heads(v)).transpose(0, 1).reshape(L, B, E)
    tgt = tgt + F.linear(o, sd[name + ".self_attn.out_proj.weight"], sd[name + ".self_attn.out_proj.bias"])
    t2 = F.layer_norm(tgt, (E,), sd[name + ".norm2.weight"], sd[name + ".norm2.bias"])
    t2 = F.linear(F.gelu(F.linear(t2, sd[name + ".linear1.weight"], sd[name + ".linear1.bias"])),
                  sd[name + ".linear2.weight"], sd[name + ".linear2.bias"])
    return tgt + t2


def calc_mean_std(feat, eps=1e-5):
    """codeformer.py:437-451: per (sample, channel) mean and sqrt(UNBIASED variance + eps)."""
    b, c = feat.shape[:2]
    var = feat.reshape(b, c, -1).var(dim=2) + eps
    return feat.reshape(b, c, -1).mean(dim=2).reshape(b, c, 1, 1), var.sqrt().reshape(b, c, 1, 1)


def adain(content, style):
    """adaptive_instance_normalization -- codeformer.py:454-470."""
    sm, ss = calc_mean_std(style)
    cm, cs = calc_mean_std(content)
    return (content - cm) / cs * ss + sm


def fuse_sft(sd, name, enc_feat, dec_feat, w):
    """Fuse_sft_block.forward -- codeformer.py:591-597."""
    enc = res_block(sd, name + ".encode_enc", torch.cat([enc_feat, dec_feat], dim=1))

    def branch(n):
        return _conv(sd, f"{name}.{n}.2", F.leaky_relu(_conv(sd, f"{name}.{n}.0", enc), 0.2))

    return dec_feat + w * (dec_feat * branch("scale") + branch("shift"))


def encode(sd, x):
    """Encoder pass of CodeFormer.forward (:694-701): returns lq_feat and the skip features by size."""
    feats = {}
    taps = {FUSE_ENCODER_BLOCK[s] for s in CONNECT}
    for i, kind in enumerate(block_kinds("encoder.blocks", sd)):
        x = run_block(sd, "encoder.blocks", i, kind, x, down=True)
        if i in taps:
            feats[str(x.shape[-1])] = x
    return x, feats


def predict_logits(sd, lq_feat):
    """Transformer + index head (:703-714): logits (B, 256, codebook)."""
    B = lq_feat.shape[0]
    pos = sd["position_emb"].unsqueeze(1).repeat(1, B, 1)
    q = F.linear(lq_feat.flatten(2).permute(2, 0, 1), sd["feat_emb.weight"], sd["feat_emb.bias"])
    for i in range(N_LAYERS):
        q = transformer_layer(sd, f"ft_layers.{i}", q, pos)
    q = F.layer_norm(q, (DIM,), sd["idx_pred_layer.0.weight"], sd["idx_pred_layer.0.bias"])
    return F.linear(q, sd["idx_pred_layer.1.weight"]).permute(1, 0, 2)


def decode(sd, top_idx, lq_feat, feats, w, use_adain):
    """Code lookup, AdaIN and generator (:727-751).  top_idx: (B, 256) int64."""
    B = top_idx.shape[0]
    quant = sd["quantize.embedding.weight"][top_idx.reshape(-1)].view(B, 16, 16, 256).permute(0, 3, 1, 2)
    if use_adain:
        quant = adain(quant, lq_feat)
    x = quant
    taps = {FUSE_GENERATOR_BLOCK[s] for s in CONNECT}
    for i, kind in enumerate(block_kinds("generator.blocks", sd)):
        x = run_block(sd, "generator.blocks", i, kind, x, down=False)
        if i in taps and w > 0:
            s = str(x.shape[-1])
            x = fuse_sft(sd, "fuse_convs_dict." + s, feats[s], x, w)
    return x


@torch.no_grad()
def codeformer_forward(sd, x, w=0.0, adain=False, code_idx=None):
    """CodeFormer.forward(x, w, adain=adain) -> (out, logits, lq_feat); x: (B, 3, 512, 512) in [-1, 1].
    ``code_idx`` (B, 256) overrides the arg-max of the logits (tests: isolates the generator from near-ties)."""
    lq_feat, feats = encode(sd, x)
    logits = predict_logits(sd, lq_feat)
    if code_idx is None:
        code_idx = torch.topk(F.softmax(logits, dim=2), 1, dim=2)[1].squeeze(2)
    out = decode(sd, code_idx, lq_feat, feats, w, adain)
    return out, logits, lq_feat
