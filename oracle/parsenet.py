"""CPU restatement of ParseNet, the face-parsing network whose class-0 mask becomes ``vsrpp_weights`` for the
bicubic tasks (scripts/video_sample.py:427-444) -- SURVEY.md section 8f row 4, parsing half.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows guided_diffusion/facelib/parsing/parsenet.py with
the configuration facelib/parsing/__init__.py:13-14 builds (``ParseNet(in_size=512, out_size=512,
parsing_ch=19)``: LeakyReLU(0.2), BatchNorm in eval mode, channels clipped to [32, 256], 4 down / 10 body / 4
up residual blocks), as functions over a STATE DICT with the reference's names.  Pinned by
tests/golden/g10_parsenet.npz (tests/test_parsenet.py).
"""
import torch
import torch.nn.functional as F


def conv_layer(sd, name, x, scale="none", act=False):
    """ConvLayer.forward -- parsenet.py:98-105: [nearest x2] -> ReflectionPad2d(1) -> Conv2d(3, stride 1 | 2) ->
    [BatchNorm2d, eval] -> [LeakyReLU(0.2)]."""
    if scale == "up":
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), sd[name + ".conv2d.weight"], sd.get(name + ".conv2d.bias"),
                 stride=2 if scale == "down" else 1)
    bn = name + ".norm.norm"
    if bn + ".weight" in sd:
        x = F.batch_norm(x, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"],
                         False, 0.0, 1e-5)
    return F.leaky_relu(x, 0.2) if act else x


def residual_block(sd, name, x, scale="none"):
    """ResidualBlock.forward -- parsenet.py:123-128; the shortcut is a plain ConvLayer when the block rescales or
    changes width (:113-116), the resampling sits on conv2 for 'down' and on conv1 for 'up' (:118-119)."""
    ident = conv_layer(sd, name + ".shortcut_func", x, scale) if name + ".shortcut_func.conv2d.weight" in sd else x
    s1, s2 = {"down": ("none", "down"), "up": ("up", "none"), "none": ("none", "none")}[scale]
    res = conv_layer(sd, name + ".conv1", x, s1, act=True)
    return ident + conv_layer(sd, name + ".conv2", res, s2)


def _count(sd, prefix):
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in sd):
        n += 1
    return n


@torch.no_grad()
def parsenet_forward(sd, x):
    """ParseNet.forward -- parsenet.py:188-194: (out_mask (B, 19, H, W), out_img (B, 3, H, W))."""
    feat = conv_layer(sd, "encoder.0", x)
    for i in range(1, _count(sd, "encoder")):
        feat = residual_block(sd, f"encoder.{i}", feat, "down")
    y = feat
    for i in range(_count(sd, "body")):
        y = residual_block(sd, f"body.{i}", y)
    y = feat + y
    for i in range(_count(sd, "decoder")):
        y = residual_block(sd, f"decoder.{i}", y, "up")
    return conv_layer(sd, "out_mask_conv", y), conv_layer(sd, "out_img_conv", y)


def face_weight(sd, frames, w_face):
    """scripts/video_sample.py:427-444: weight = mask * w_face + (1 - mask), mask = (argmax of the parsing logits == 0)."""
    mask = (parsenet_forward(sd, frames)[0].argmax(1, keepdim=True) == 0).float()
    return mask * w_face + (1 - mask) * 1.0
