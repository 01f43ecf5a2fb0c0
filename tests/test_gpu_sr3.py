"""sr3.UNet (bicubic tasks) on the GPU vs the CPU oracle; plus the antialiased resize helper."""
import pytest
import torch
import torch.nn.functional as F

from tests.golden.weights import name_seeded_weights

SR3_SMALL = dict(image_size=64, in_channel=6, out_channel=3, inner_channel=64, norm_groups=16,
                 channel_mults=(1, 2, 4), attn_res=(32, 16), vsrpp_res=(64,), spatial_attn=False,
                 temporal_attn=True, res_blocks=1, dropout=0.0, dtype=torch.float32, cross_frame_module=True,
                 use_checkpoint=False, num_frames=7, head_dim=64)


def build_pair():
    from oracle.sr3 import UNet as Oracle
    from flair_amd.guided_diffusion.sr3 import UNet
    torch.manual_seed(0)
    o = name_seeded_weights(Oracle(**SR3_SMALL)).eval()
    m = UNet(**SR3_SMALL)
    m.load_state_dict(o.state_dict(), strict=True)
    return o, m.eval()


def test_sr3_state_dict_names_match_oracle():
    o, m = build_pair()
    assert list(o.state_dict().keys()) == list(m.state_dict().keys())


def inputs(T=4, S=64, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, 3, S, S, generator=g)
    base = torch.rand(3, S, S, generator=g) * 2 - 1
    lr = torch.stack([torch.roll(base, shifts=(i, 2 * i), dims=(1, 2)) for i in range(T)])[None]
    lr = (lr + 0.05 * torch.randn(1, T, 3, S, S, generator=g)).clamp(-1, 1)
    level = torch.full((T,), 0.83)
    return x, lr, level


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sr3_small_vs_oracle(dev, dtype):
    o, m = build_pair()
    x, lr, level = inputs()
    with torch.no_grad():
        ref = o(x, level, low_res_input=lr, num_frames=4, vsrpp_weights=0.93)
    m = m.to(dev)
    if dtype == torch.bfloat16:
        m.convert_to_fp16()
    y = m(x.to(dev), level.to(dev), low_res_input=lr.to(dev), num_frames=4, vsrpp_weights=0.93)
    torch.cuda.synchronize()
    rel = 3e-4 if dtype == torch.float32 else 5e-2
    err = (y.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= rel, err


@pytest.mark.gpu
@pytest.mark.parametrize("T", [1, 2, 9])
def test_sr3_edge_frame_counts_vs_oracle(dev, T):
    """One frame (nothing to propagate), two frames (first-order alignment only), and more frames than the
    temporal attention was configured for (num_frames=7): f32 kernels, tensor-valued vsrpp weights."""
    o, m = build_pair()
    x, lr, level = inputs(T=T, seed=9)
    g = torch.Generator().manual_seed(T)
    vw = 0.9 + 0.1 * torch.rand(1, T, 1, 64, 64, generator=g)
    with torch.no_grad():
        ref = o(x, level, low_res_input=lr, num_frames=T, vsrpp_weights=vw)
    m = m.to(dev)
    y = m(x.to(dev), level.to(dev), low_res_input=lr.to(dev), num_frames=T, vsrpp_weights=vw.to(dev))
    torch.cuda.synchronize()
    err = (y.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 3e-4, err


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(96, 64), (256, 128), (100, 64)])
def test_antialiased_resize(dev, size):
    from flair_amd.guided_diffusion.sr3 import aa_resize
    n_in, n_out = size
    g = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, n_in, n_in, generator=g) * 2 - 1
    ref = F.interpolate(x, size=(n_out, n_out), mode="bilinear", align_corners=False, antialias=True)
    y = aa_resize(x.to(dev), n_out).cpu()
    assert (y - ref).abs().max().item() <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_strided_conv_and_frame_bias(dev, dtype):
    import math
    from flair_amd import ops
    from tests.util import assert_close, from_clip, rb, to_clip
    g = torch.Generator().manual_seed(2)
    T, H, W, cin, cout = 3, 18, 14, 64, 96
    x = rb(torch.randn(T, cin, H, W, generator=g), dtype)
    w = rb(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin), dtype)
    b = torch.randn(cout, generator=g) * 0.1
    fb = torch.randn(T, cout + 7, generator=g)
    ref = F.conv2d(x, w, b, stride=2, padding=1) + fb[:, :cout, None, None]
    wp = ops.pack_conv_weight(w, [(cin, cin)], dtype).to(dev)
    y = ops.conv(to_clip(x, dtype, dev), wp, b.to(dev), cout, (1, 3, 3), stride=2, frame_bias=fb.to(dev)[:, :cout])
    torch.cuda.synchronize()
    assert_close(from_clip(y), ref, dtype, "stride-2 conv + frame bias")


@pytest.mark.gpu
def test_gated_blend_and_sin_first_encoding(dev):
    from flair_amd import ops
    from oracle.sr3 import PositionalEncoding
    from tests.util import from_clip, to_clip
    g = torch.Generator().manual_seed(3)
    x, m = torch.randn(3, 64, 5, 6, generator=g), torch.randn(3, 64, 5, 6, generator=g)
    gate = torch.randn(3, 70, generator=g)
    s = torch.sigmoid(gate[:, :64])[:, :, None, None]
    ref = (1 - s) * x + s * m
    y = ops.gated_blend(to_clip(x, torch.float32, dev), to_clip(m, torch.float32, dev), gate.to(dev)[:, :64])
    assert (from_clip(y) - ref).abs().max().item() <= 1e-5
    lv = torch.tensor([0.1, 0.83, 1.0])
    assert (ops.timestep_embedding(lv.to(dev), 64, sin_first=True).cpu() - PositionalEncoding(64)(lv)).abs().max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tensor_weights", [False, True])
def test_sr3_hip_graph_replay_matches_eager(dev, tensor_weights):
    """sr3.UNet.enable_hip_graph(): the captured forward replays bit-identically to eager launches for new latents /
    noise levels, after the conditioning clip changes, and with the per-pixel propagation-weight map of the bicubic
    tasks (a tensor input of the graph)."""
    _, m = build_pair()
    m = m.to(dev)
    m.convert_to_fp16()
    T, S = 4, 64
    g = torch.Generator().manual_seed(2)
    wmap = (0.9 + 0.1 * torch.rand(1, T, 1, S, S, generator=g)).to(dev) if tensor_weights else 0.93
    cases, clips = [], {}
    for seed, lvl in [(5, 0.83), (5, 0.31), (5, 0.55), (9, 0.97)]:   # the same clip three times, then a new clip
        x, lr, level = inputs(T, S, seed=seed)
        if seed not in clips:
            clips[seed] = lr.to(dev)                           # ONE device tensor per clip: later calls are pure replays
        cases.append(((x + lvl).to(dev), clips[seed], torch.full((T,), lvl).to(dev)))
    eager = [m(x, lv, low_res_input=lr, num_frames=T, vsrpp_weights=wmap).clone() for x, lr, lv in cases]
    m.enable_hip_graph()
    graphs = []
    for (x, lr, lv), ref in zip(cases, eager):
        y = m(x, lv, low_res_input=lr, num_frames=T, vsrpp_weights=wmap)
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
        assert len(m._graphs) == 1
        graphs.append(next(iter(m._graphs.values()))["graph"])
    assert graphs[0] is graphs[1] is graphs[2] and graphs[3] is not graphs[0]   # replays with new x / level, then a re-capture
    m.enable_hip_graph(False)
