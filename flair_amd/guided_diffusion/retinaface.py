"""RetinaFace face / landmark detector on the HIP kernels (detection half).

Mirror of ``guided_diffusion/facelib/detection/retinaface/retinaface.py:79-418`` (``RetinaFace(network_name, half, phase,
device)``, ``forward``, ``detect_faces``-family; the sampling script reaches ``batched_detect_faces`` through
``FaceRestoreHelper.get_crop_face``, ``facelib/utils/face_restoration_helper.py:122-224``, once per window) and of the
blocks in ``retinaface_net.py:1-196`` (``FPN``, ``SSH``, ``ClassHead`` / ``BboxHead`` / ``LandmarkHead``).  The body is
torchvision's ResNet-50 behind ``IntermediateLayerGetter`` (``retinaface.py:99-102``: conv1 / bn1 / maxpool / layer1-4,
outputs of layer2, layer3, layer4) -- torchvision is not importable here, so its v1.5 Bottleneck layout (stride on the 3x3
convolution) is restated and that part is PARITY UNPINNED; the state-dict names are the reference's
(``detection_Resnet50_Final.pth`` with its ``module.`` prefixes stripped as ``facelib/detection/__init__.py:39-43`` does
loads unchanged).

The ``nn.Module`` classes are parameter containers.  Every Conv2d + eval BatchNorm (+ LeakyReLU / ReLU) is ONE
``flair_conv_nhwc`` launch with the BatchNorm folded into the packed weights; the Bottleneck's ``relu(out + identity)`` and
FPN's lateral sums are ``flair_add_act_nhwc``; SSH's ``relu(cat(...))`` is the three branch convolutions writing their
channel slices of one tensor with the ReLU in their epilogues; the heads' ``permute(0, 2, 3, 1)`` is the NHWC layout itself.
Priors, decoding, NMS run on the host in numpy (``retinaface_utils.py``; the reference moves the decoded boxes to the host at
once, too).  float32 throughout, like the reference (``half=False``).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib, ops
from .. import ops as A
from .retinaface_utils import PriorBox, batched_decode, batched_decode_landm, decode, decode_landm, py_cpu_nms


def generate_config(network_name):
    """retinaface.py:32-77 (the inference-relevant keys)."""
    if network_name == "resnet50":
        return {"name": "Resnet50", "min_sizes": [[16, 32], [64, 128], [256, 512]], "steps": [8, 16, 32], "variance": [0.1, 0.2],
                "clip": False, "return_layers": {"layer2": 1, "layer3": 2, "layer4": 3}, "in_channel": 256, "out_channel": 256}
    if network_name == "mobile0.25":
        raise NotImplementedError("flair_amd: the MobileNet-0.25 body (depthwise convolutions) is not built; FLAIR uses "
                                  "det_model='retinaface_resnet50' (face_restoration_helper.py:69)")
    raise NotImplementedError(f"network_name={network_name}")


class _ConvBN(nn.Sequential):
    """conv_bn / conv_bn_no_relu / conv_bn1X1 of retinaface_net.py:6-24: Sequential(Conv2d(bias=False), BatchNorm2d[, LeakyReLU])."""

    def __init__(self, inp, oup, k, stride, leaky=None):
        layers = [nn.Conv2d(inp, oup, k, stride, k // 2, bias=False), nn.BatchNorm2d(oup)]
        if leaky is not None:
            layers.append(nn.LeakyReLU(negative_slope=leaky, inplace=True))
        super().__init__(*layers)
        self.leaky = leaky

    def pack(self, dtype, device, relu_after=False):
        self._p = _fold(self[0], self[1], dtype, device)
        lk = self.leaky
        self._act = (A.ACT_RELU if lk == 0 else A.ACT_LRELU01) if lk is not None else (A.ACT_RELU if relu_after else A.ACT_NONE)
        assert lk in (None, 0, 0.1)

    def run(self, x, out=None):
        return _conv(x, self._p, self[0], self._act, out=out)


def _fold(conv, bn, dtype, device):
    """Packed weights / bias of conv followed by an eval-mode BatchNorm (bn may be None)."""
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else w.new_zeros(w.shape[0])
    if bn is not None:
        g = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = w * g.view(-1, 1, 1, 1)
        b = (b - bn.running_mean.detach().float()) * g + bn.bias.detach().float()
    cout, cin = w.shape[0], w.shape[1]
    cpad = (cout + 3) // 4 * 4
    wp = ops.pack_conv_weight(w.to(device), [(cin, ops.pad_channels(cin, dtype))], dtype, cpad)
    b = b.to(device)
    return wp, torch.cat([b, b.new_zeros(cpad - cout)]).contiguous(), cpad


def _conv(x, packed, conv, act, out=None, res0=None):
    wp, b, cout = packed
    k = conv.kernel_size[0]
    return ops.conv(x, wp, b, cout, (1, k, k), stride=conv.stride[0], act=act, out=out, res0=res0)


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5: the stride sits on conv2), expansion 4."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        if downsample is not None:
            self.downsample = downsample

    def pack(self, dtype, device):
        self._p = [_fold(self.conv1, self.bn1, dtype, device), _fold(self.conv2, self.bn2, dtype, device),
                   _fold(self.conv3, self.bn3, dtype, device)]
        self._pd = _fold(self.downsample[0], self.downsample[1], dtype, device) if hasattr(self, "downsample") else None

    def run(self, x):
        ident = _conv(x, self._pd, self.downsample[0], A.ACT_NONE) if self._pd is not None else x
        h = _conv(x, self._p[0], self.conv1, A.ACT_RELU)
        h = _conv(h, self._p[1], self.conv2, A.ACT_RELU)
        h = _conv(h, self._p[2], self.conv3, A.ACT_NONE)
        return ops.add_act(h, ident, A.ACT_RELU, out=h)          # out += identity; relu


class _ResNet50Body(nn.Module):
    """What IntermediateLayerGetter(resnet50, {'layer2': 1, 'layer3': 2, 'layer4': 3}) keeps: conv1, bn1, relu, maxpool,
    layer1 .. layer4 (avgpool / fc are dropped, so the state dict has no fc.*)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for i, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
            down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
            layers = [Bottleneck(inplanes, planes, stride, down)]
            inplanes = planes * 4
            layers += [Bottleneck(inplanes, planes) for _ in range(1, blocks)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*layers))

    def pack(self, dtype, device):
        self._p = _fold(self.conv1, self.bn1, dtype, device)
        for m in self.modules():
            if isinstance(m, Bottleneck):
                m.pack(dtype, device)

    def run(self, x):
        h = ops.maxpool3x3s2(_conv(x, self._p, self.conv1, A.ACT_RELU))
        outs = []
        for i in range(1, 5):
            for blk in getattr(self, f"layer{i}"):
                h = blk.run(h)
            if i >= 2:
                outs.append(h)
        return outs


class SSH(nn.Module):
    """retinaface_net.py:37-63."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        assert out_channel % 4 == 0
        leaky = 0.1 if out_channel <= 64 else 0
        self.conv3X3 = _ConvBN(in_channel, out_channel // 2, 3, 1)
        self.conv5X5_1 = _ConvBN(in_channel, out_channel // 4, 3, 1, leaky)
        self.conv5X5_2 = _ConvBN(out_channel // 4, out_channel // 4, 3, 1)
        self.conv7X7_2 = _ConvBN(out_channel // 4, out_channel // 4, 3, 1, leaky)
        self.conv7x7_3 = _ConvBN(out_channel // 4, out_channel // 4, 3, 1)
        self.out_channel = out_channel

    def pack(self, dtype, device):
        for m, relu_after in ((self.conv3X3, True), (self.conv5X5_1, False), (self.conv5X5_2, True), (self.conv7X7_2, False),
                              (self.conv7x7_3, True)):
            m.pack(dtype, device, relu_after)                   # F.relu(cat(...)) = the ReLU in each branch's epilogue

    def run(self, x):
        T, H, W, _ = x.shape
        c = self.out_channel
        out = torch.empty((T, H, W, c), dtype=x.dtype, device=x.device)
        self.conv3X3.run(x, out=out[..., :c // 2])
        t = self.conv5X5_1.run(x)
        self.conv5X5_2.run(t, out=out[..., c // 2:3 * c // 4])
        self.conv7x7_3.run(self.conv7X7_2.run(t), out=out[..., 3 * c // 4:])
        return out


class FPN(nn.Module):
    """retinaface_net.py:66-97."""

    def __init__(self, in_channels_list, out_channels):
        super().__init__()
        leaky = 0.1 if out_channels <= 64 else 0
        self.output1 = _ConvBN(in_channels_list[0], out_channels, 1, 1, leaky)
        self.output2 = _ConvBN(in_channels_list[1], out_channels, 1, 1, leaky)
        self.output3 = _ConvBN(in_channels_list[2], out_channels, 1, 1, leaky)
        self.merge1 = _ConvBN(out_channels, out_channels, 3, 1, leaky)
        self.merge2 = _ConvBN(out_channels, out_channels, 3, 1, leaky)

    def pack(self, dtype, device):
        for m in (self.output1, self.output2, self.output3, self.merge1, self.merge2):
            m.pack(dtype, device)

    def run(self, feats):
        o1, o2, o3 = self.output1.run(feats[0]), self.output2.run(feats[1]), self.output3.run(feats[2])
        up3 = ops.resize(o3, (o2.shape[1], o2.shape[2]), ops.RESIZE_NEAREST)
        o2 = self.merge2.run(ops.add_act(o2, up3, A.ACT_NONE, out=up3))
        up2 = ops.resize(o2, (o1.shape[1], o1.shape[2]), ops.RESIZE_NEAREST)
        o1 = self.merge1.run(ops.add_act(o1, up2, A.ACT_NONE, out=up2))
        return [o1, o2, o3]


class _Head(nn.Module):
    """ClassHead / BboxHead / LandmarkHead of retinaface_net.py:139-178: a biased 1x1 convolution, then
    permute(0, 2, 3, 1).view(B, -1, k) -- which the NHWC result already is."""

    def __init__(self, inchannels, num_anchors, k):
        super().__init__()
        self.k = k
        self.conv1x1 = nn.Conv2d(inchannels, num_anchors * k, kernel_size=(1, 1), stride=1, padding=0)

    def pack(self, dtype, device):
        self._p = _fold(self.conv1x1, None, dtype, device)

    def run(self, x):
        y = _conv(x, self._p, self.conv1x1, A.ACT_NONE)
        return y[..., :self.conv1x1.out_channels].reshape(x.shape[0], -1, self.k)


class ClassHead(_Head):
    def __init__(self, inchannels=512, num_anchors=3):
        super().__init__(inchannels, num_anchors, 2)


class BboxHead(_Head):
    def __init__(self, inchannels=512, num_anchors=3):
        super().__init__(inchannels, num_anchors, 4)


class LandmarkHead(_Head):
    def __init__(self, inchannels=512, num_anchors=3):
        super().__init__(inchannels, num_anchors, 10)


def make_class_head(fpn_num=3, inchannels=64, anchor_num=2):
    return nn.ModuleList([ClassHead(inchannels, anchor_num) for _ in range(fpn_num)])


def make_bbox_head(fpn_num=3, inchannels=64, anchor_num=2):
    return nn.ModuleList([BboxHead(inchannels, anchor_num) for _ in range(fpn_num)])


def make_landmark_head(fpn_num=3, inchannels=64, anchor_num=2):
    return nn.ModuleList([LandmarkHead(inchannels, anchor_num) for _ in range(fpn_num)])


class RetinaFace(nn.Module):
    """retinaface.py:79-418.  ``forward`` takes (B, 3, H, W) mean-subtracted float images (as the reference's does) and
    returns (bbox_regressions (B, N, 4), softmax(classifications) (B, N, 2), ldm_regressions (B, N, 10)) on the device."""

    def __init__(self, network_name="resnet50", half=False, phase="test", device="cuda"):
        super().__init__()
        if half:
            raise NotImplementedError("flair_amd: the detector runs in float32 (the reference's helper builds it with half=False)")
        cfg = generate_config(network_name)
        self.half_inference = False
        self.backbone = cfg["name"]
        self.device = torch.device(device)
        self.model_name = f"retinaface_{network_name}"
        self.cfg = cfg
        self.phase = phase
        self.target_size, self.max_size = 1600, 2150
        self.resize, self.scale, self.scale1 = 1.0, None, None
        self.mean = (104.0, 117.0, 123.0)
        self.body = _ResNet50Body()
        c2 = cfg["in_channel"]
        oc = cfg["out_channel"]
        self.fpn = FPN([c2 * 2, c2 * 4, c2 * 8], oc)
        self.ssh1, self.ssh2, self.ssh3 = SSH(oc, oc), SSH(oc, oc), SSH(oc, oc)
        self.ClassHead = make_class_head(fpn_num=3, inchannels=oc)
        self.BboxHead = make_bbox_head(fpn_num=3, inchannels=oc)
        self.LandmarkHead = make_landmark_head(fpn_num=3, inchannels=oc)
        self.dtype = torch.float32
        self._packed_key = None
        self.to(self.device)
        self.eval()

    def load_state_dict(self, state_dict, *args, **kwargs):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}   # detection/__init__.py:39-43
        out = super().load_state_dict(sd, *args, **kwargs)
        self._packed_key = None
        return out

    def _ensure_packed(self, device):
        key = (self.dtype, device)
        if self._packed_key != key:
            self.body.pack(self.dtype, device)
            self.fpn.pack(self.dtype, device)
            for m in (self.ssh1, self.ssh2, self.ssh3, *self.ClassHead, *self.BboxHead, *self.LandmarkHead):
                m.pack(self.dtype, device)
            self._packed_key = key

    def _to_clip(self, inputs):
        if not inputs.is_cuda:
            raise _lib.FlairHipError("flair_amd RetinaFace needs its input in HBM (device='cuda'); no CPU path exists")
        self._ensure_packed(inputs.device)
        B, _, H, W = inputs.shape
        x = torch.zeros((B, H, W, ops.pad_channels(3, self.dtype)), dtype=self.dtype, device=inputs.device)
        return ops.nchw_to_clip(inputs.float().contiguous(), x, 0)

    def _run_clip(self, x):
        """(B, H, W, >= 3) float32 clip tensor, mean already subtracted -> raw head outputs on the device."""
        return self._neck_heads(self.body.run(x))

    def _neck_heads(self, body_feats):
        """FPN -> SSH -> heads on the three body outputs (clip tensors of 512 / 1024 / 2048 channels)."""
        fpn = self.fpn.run(body_feats)
        feats = [self.ssh1.run(fpn[0]), self.ssh2.run(fpn[1]), self.ssh3.run(fpn[2])]
        bbox = torch.cat([self.BboxHead[i].run(f) for i, f in enumerate(feats)], dim=1)
        cls = torch.cat([self.ClassHead[i].run(f) for i, f in enumerate(feats)], dim=1)
        ldm = torch.cat([self.LandmarkHead[i].run(f) for i, f in enumerate(feats)], dim=1)
        return bbox, cls, ldm

    @torch.no_grad()
    def forward(self, inputs):
        bbox, cls, ldm = self._run_clip(self._to_clip(inputs))
        if self.phase == "train":
            return bbox, cls, ldm
        # (B, N, 2) softmax of the API edge; the detect_* methods below evaluate it on the host with the decoding
        return bbox, torch.softmax(cls, dim=-1), ldm

    # -- detection (retinaface.py:157-262, 294-418): the network on the GPU, priors / decoding / NMS on the host
    @torch.no_grad()
    def _detect_raw(self, frames, pre=None):
        """frames: (B, 3, H, W) float, mean NOT yet subtracted -> host arrays (loc, score of class 1, landmarks, priors).
        pre = (a, b, lo, hi): the frames are first mapped to clamp(a x + b, lo, hi) (the helper's [-1, 1] -> [0, 255])."""
        B, _, H, W = frames.shape
        x = self._to_clip(frames.to(self.device))
        sub = torch.tensor(self.mean, dtype=torch.float32, device=x.device)            # image - self.mean_tensor
        a_, b_, lo_, hi_ = pre if pre is not None else (1.0, 0.0, float("-inf"), float("inf"))
        ops.affine_channels(x, 3, a_, b_, lo_, hi_, sub, torch.ones(3, dtype=torch.float32, device=x.device), x)
        loc, cls, ldm = self._run_clip(x)
        loc, cls, ldm = loc.cpu().numpy(), cls.cpu().numpy().astype(np.float32), ldm.cpu().numpy()
        m = cls.max(axis=-1, keepdims=True)                       # F.softmax(classifications, dim=-1)[..., 1]
        e = np.exp(cls - m)
        conf = e[..., 1] / e.sum(axis=-1)
        self.scale = np.array([W, H, W, H], dtype=np.float32)
        self.scale1 = np.array([W, H] * 5, dtype=np.float32)
        priors = PriorBox(self.cfg, image_size=(H, W)).forward()
        return loc, conf, ldm, priors

    def detect_faces(self, image, conf_threshold=0.8, nms_threshold=0.4, use_origin_size=True):
        """image: (H, W, 3) BGR array or (1, 3, H, W) tensor in [0, 255].  Returns (n, 15): box, score, 5 landmarks."""
        if not use_origin_size:
            raise NotImplementedError("flair_amd: use_origin_size=False (cv2.resize of the input) is not built")
        if not torch.is_tensor(image):
            image = torch.from_numpy(np.asarray(image, dtype=np.float32).transpose(2, 0, 1)[None]).to(self.device)
        self.resize = 1
        loc, conf, ldm, priors = self._detect_raw(image.to(self.device))
        boxes = decode(loc[0], priors, self.cfg["variance"]) * self.scale / self.resize
        scores = conf[0]
        landmarks = decode_landm(ldm[0], priors, self.cfg["variance"]) * self.scale1 / self.resize
        inds = np.where(scores > conf_threshold)[0]
        boxes, landmarks, scores = boxes[inds], landmarks[inds], scores[inds]
        order = scores.argsort()[::-1]
        boxes, landmarks, scores = boxes[order], landmarks[order], scores[order]
        dets = np.hstack((boxes, scores[:, np.newaxis])).astype(np.float32, copy=False)
        keep = py_cpu_nms(dets, nms_threshold)
        return np.concatenate((dets[keep, :], landmarks[keep]), axis=1)

    @torch.no_grad()
    def batched_detect_faces(self, frames, conf_threshold=0.8, nms_threshold=0.4, use_origin_size=True, pre=None):
        """frames: (B, 3, H, W) float tensor in [0, 255] (what FaceRestoreHelper.get_crop_face passes).  Returns a list with
        one (n_i, 15) array per frame that has detections (frames without are skipped, retinaface.py:393-395)."""
        if not use_origin_size:
            raise NotImplementedError("flair_amd: use_origin_size=False is not built")
        self.resize = 1
        b_loc, b_conf, b_ldm, priors = self._detect_raw(frames.to(self.device), pre)
        priors = priors[None]
        b_loc = batched_decode(b_loc, priors, self.cfg["variance"]) * self.scale / self.resize
        b_ldm = batched_decode_landm(b_ldm, priors, self.cfg["variance"]) * self.scale1 / self.resize
        final = []
        for loc, conf, ldm in zip(b_loc, b_conf, b_ldm):
            inds = conf > conf_threshold
            if not inds.any():
                continue
            dets = np.concatenate((loc[inds], conf[inds, None]), axis=1).astype(np.float32)
            ldm = ldm[inds]
            keep = py_cpu_nms(dets, nms_threshold)
            final.append(np.concatenate((dets[keep, :], ldm[keep]), axis=1))
        return final
