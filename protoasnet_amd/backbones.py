"""Feature trunks: parameter containers with the reference's ``state_dict`` keys, HIP forward.

Each trunk is an ``nn.Module`` whose sub-module names reproduce the checkpoint keys
of the reference (``cnn_backbone.conv1.weight``, ``cnn_backbone.backbone.1.0.conv1.0.0.weight``
...; SURVEY.md section 8b "Checkpoint keys"), so reference checkpoints load with
``load_state_dict``.  The sub-modules only *hold* parameters: ``forward`` compiles the
trunk once into a list of fused HIP launches (``protoasnet_amd.plan``) -- norm layers are
folded into per-channel scale/bias epilogues, activations live in channels-last bf16/fp32
buffers -- and replays it.  There is no torch / CPU fallback.

* ``resnet18_features``  -- reference src/models/resnet_features.py:126-248 (2-D trunk).
* ``resnet2p1d_18``      -- reference src/models/resnet_features.py:307-327
  (torchvision ``r2plus1d_18`` children[:last_layer_num]).
* ``x3d_s`` / ``x3d_m``  -- not in the reference; named by BASELINE.json (definition in DESIGN.md).
"""
from __future__ import annotations

import os

from typing import List, Tuple

import torch
import torch.nn as nn

from . import plan as _plan


# ----------------------------------------------------------------------------- pretrained weight files
# The reference fetches these with ``model_zoo.load_url(url, model_dir="./pretrained_models")`` (resnet_features.py:8-18): the file is
# cached as ``<model_dir>/<basename of the url>`` and read from there on every later run.  This build has no network, so the cache file
# IS the interface: drop the file the reference would have downloaded into ``./pretrained_models`` (relative to the working directory,
# like the reference) and ``pretrained: True`` configs build unchanged; a missing file is an error that names it.
model_urls = {
    "resnet18": "https://download.pytorch.org/models/resnet18-5c106cde.pth",
    "resnet2p1d_18": "https://download.pytorch.org/models/r2plus1d_18-91a641e6.pth",
}
model_dir = "./pretrained_models"


def load_pretrained(name: str) -> dict:
    url = model_urls[name]
    cached = os.path.join(model_dir, os.path.basename(url))
    if not os.path.exists(cached):
        raise FileNotFoundError(
            f"pretrained=True: {cached} not found.  The reference downloads {url} into {model_dir}/ (resnet_features.py:8-18); this "
            f"build never touches the network -- place that file there, or construct with pretrained=False and load_state_dict() a "
            f"checkpoint"
        )
    return torch.load(cached, map_location="cpu")


def _r2plus1d_keys_to_backbone(sd: dict, n_children: int) -> dict:
    """torchvision ``r2plus1d_18`` keys (``stem.*``, ``layerK.*``, ``fc.*``) -> the ``backbone.<child index>.*`` keys of the slice."""
    out = {}
    for k, v in sd.items():
        head, _, rest = k.partition(".")
        idx = 0 if head == "stem" else int(head[5:]) if head.startswith("layer") and head[5:].isdigit() else -1
        if 0 <= idx < n_children:
            out[f"backbone.{idx}.{rest}"] = v
    return out


# ----------------------------------------------------------------------------- 2-D ResNet-18
class _BasicBlock2d(nn.Module):
    def __init__(self, inplanes: int, planes: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride


class ResNet18Features(_plan.HipTrunk):
    """conv7x7/2 - BN - ReLU - maxpool3x3/2 - 4 stages x 2 BasicBlocks; output (N, 512, H/32, W/32)."""

    arch = "resnet18"
    out_channels = 512

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        self._conv_info: Tuple[List[int], List[int], List[int]] = ([7, 3], [2, 2], [3, 1])
        for li, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), start=1):
            blocks = []
            for b in range(2):
                s = stride if b == 0 else 1
                blocks.append(_BasicBlock2d(inplanes, planes, s))
                inplanes = planes
                self._conv_info[0].extend([3, 3])
                self._conv_info[1].extend([s, 1])
                self._conv_info[2].extend([1, 1])
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        for m in self.modules():  # resnet_features.py:157-162
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def conv_info(self):
        return self._conv_info

    def __repr__(self):
        return "resnet18_features"

    def build_plan(self, pb: "_plan.PlanBuilder", x: "_plan.Act") -> "_plan.Act":
        x = pb.first_conv(x, self.conv1, self.bn1, act="relu")
        x = pb.maxpool(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}"):
                identity = x
                if blk.downsample is not None:
                    identity = pb.conv(x, blk.downsample[0], blk.downsample[1], act="none")
                y = pb.conv(x, blk.conv1, blk.bn1, act="relu")
                x = pb.conv(y, blk.conv2, blk.bn2, act="relu", residual=identity)
        return x


    def build_train(self, tb, x):
        x = tb.unit(x, self.conv1, self.bn1, "relu", kind="first")
        x = tb.maxpool_unit(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}"):
                identity = x if blk.downsample is None else tb.unit(x, blk.downsample[0], blk.downsample[1], "none")
                y = tb.unit(x, blk.conv1, blk.bn1, "relu")
                x = tb.unit(y, blk.conv2, blk.bn2, "relu", residual=identity)
        return x


def resnet18_features(pretrained: bool = False, **kwargs) -> ResNet18Features:
    """resnet_features.py:236-248: ``pretrained`` loads the ImageNet file from ``model_dir`` with ``fc.*`` popped, ``strict=False``."""
    model = ResNet18Features()
    if pretrained:
        my_dict = load_pretrained("resnet18")
        my_dict.pop("fc.weight")  # KeyError when absent, like the reference
        my_dict.pop("fc.bias")
        model.load_state_dict(my_dict, strict=False)
    return model


# ----------------------------------------------------------------------------- R(2+1)D-18
def _midplanes(inplanes: int, planes: int) -> int:
    return (inplanes * planes * 27) // (inplanes * 9 + 3 * planes)


class _Conv2Plus1D(nn.Sequential):
    def __init__(self, inplanes: int, planes: int, mid: int, stride: int = 1):
        super().__init__(
            nn.Conv3d(inplanes, mid, (1, 3, 3), (1, stride, stride), (0, 1, 1), bias=False),
            nn.BatchNorm3d(mid),
            nn.ReLU(inplace=True),
            nn.Conv3d(mid, planes, (3, 1, 1), (stride, 1, 1), (1, 0, 0), bias=False),
        )


class _BasicBlock2p1d(nn.Module):
    def __init__(self, inplanes: int, planes: int, stride: int):
        super().__init__()
        mid = _midplanes(inplanes, planes)
        self.conv1 = nn.Sequential(_Conv2Plus1D(inplanes, planes, mid, stride), nn.BatchNorm3d(planes), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(_Conv2Plus1D(planes, planes, mid), nn.BatchNorm3d(planes))
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(
                nn.Conv3d(inplanes, planes, 1, (stride, stride, stride), bias=False), nn.BatchNorm3d(planes)
            )


class resnet2p1d_18(_plan.HipTrunk):  # noqa: N801 -- name is part of the reference surface (ProtoPNet.py:36)
    """``children(r2plus1d_18)[:last_layer_num]`` under ``self.backbone`` (-3: 256 ch, T/4, H/8, W/8)."""

    arch = "resnet2p1d_18"

    def __init__(self, pretrained: bool = True, last_layer_num: int = -3, **kwargs):
        super().__init__()
        n_children = 7 + last_layer_num
        if not 1 <= n_children <= 5:
            raise ValueError("last_layer_num must keep between 1 and 5 convolutional children (-6..-2)")
        stem = nn.Sequential(
            nn.Conv3d(3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), bias=False), nn.BatchNorm3d(45), nn.ReLU(inplace=True),
            nn.Conv3d(45, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), bias=False), nn.BatchNorm3d(64), nn.ReLU(inplace=True),
        )
        children = [stem]
        inplanes = 64
        for li in range(1, n_children):
            planes, stride = 64 * 2 ** (li - 1), (1 if li == 1 else 2)
            children.append(nn.Sequential(_BasicBlock2p1d(inplanes, planes, stride), _BasicBlock2p1d(planes, planes, 1)))
            inplanes = planes
        self.backbone = nn.Sequential(*children)
        self.out_channels = inplanes
        if pretrained:
            # resnet_features.py:316-319 loads the Kinetics file into the WHOLE torchvision model (strict=False) and then keeps
            # children[:last_layer_num]: stem -> backbone.0, layerK -> backbone.K; fc.* and the layers beyond the cut are dropped with it.
            self.load_state_dict(_r2plus1d_keys_to_backbone(load_pretrained("resnet2p1d_18"), n_children), strict=False)

    def __repr__(self):
        return f"resnet2p1d_18(children={len(self.backbone)}, out_channels={self.out_channels})"

    def build_plan(self, pb, x):
        stem = self.backbone[0]
        x = pb.first_conv(x, stem[0], stem[1], act="relu")
        x = pb.conv(x, stem[3], stem[4], act="relu")
        for layer in list(self.backbone)[1:]:
            for blk in layer:
                identity = x
                if blk.downsample is not None:
                    identity = pb.conv(x, blk.downsample[0], blk.downsample[1], act="none")
                c = blk.conv1[0]
                y = pb.conv(x, c[0], c[1], act="relu")
                y = pb.conv(y, c[3], blk.conv1[1], act="relu")
                c = blk.conv2[0]
                y = pb.conv(y, c[0], c[1], act="relu")
                x = pb.conv(y, c[3], blk.conv2[1], act="relu", residual=identity)
        return x


    def build_train(self, tb, x):
        stem = self.backbone[0]
        x = tb.unit(x, stem[0], stem[1], "relu", kind="first")
        x = tb.unit(x, stem[3], stem[4], "relu")
        for layer in list(self.backbone)[1:]:
            for blk in layer:
                identity = x if blk.downsample is None else tb.unit(x, blk.downsample[0], blk.downsample[1], "none")
                c = blk.conv1[0]
                y = tb.unit(x, c[0], c[1], "relu")
                y = tb.unit(y, c[3], blk.conv1[1], "relu")
                c = blk.conv2[0]
                y = tb.unit(y, c[0], c[1], "relu")
                x = tb.unit(y, c[3], blk.conv2[1], "relu", residual=identity)
        return x


# ----------------------------------------------------------------------------- X3D
X3D_STEM_DIM = 24
X3D_STAGES = ((24, 3), (48, 5), (96, 11), (192, 7))
X3D_BOTTLENECK = 2.25
X3D_SE_RATIO = 0.0625


def _round_width(width: float, multiplier: float, min_width: int = 8, divisor: int = 8) -> int:
    width *= multiplier
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


class _X3DStem(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.conv_xy = nn.Conv3d(3, dim, (1, 3, 3), (1, 2, 2), (0, 1, 1), bias=False)
        self.conv_t = nn.Conv3d(dim, dim, (5, 1, 1), 1, (2, 0, 0), groups=dim, bias=False)
        self.bn = nn.BatchNorm3d(dim)


class _SE(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        mid = _round_width(dim, X3D_SE_RATIO)
        self.fc1 = nn.Conv3d(dim, mid, 1)
        self.fc2 = nn.Conv3d(mid, dim, 1)


class _ShortCut(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, 1, (1, stride, stride), bias=False)
        self.bn = nn.BatchNorm3d(cout)


class _X3DBlock(nn.Module):
    def __init__(self, dim_in: int, dim_out: int, stride: int, use_se: bool):
        super().__init__()
        inner = int(X3D_BOTTLENECK * dim_out)
        self.conv_a = nn.Conv3d(dim_in, inner, 1, bias=False)
        self.bn_a = nn.BatchNorm3d(inner)
        self.conv_b = nn.Conv3d(inner, inner, 3, (1, stride, stride), 1, groups=inner, bias=False)
        self.bn_b = nn.BatchNorm3d(inner)
        self.se = _SE(inner) if use_se else None
        self.conv_c = nn.Conv3d(inner, dim_out, 1, bias=False)
        self.bn_c = nn.BatchNorm3d(dim_out)
        self.shortcut = _ShortCut(dim_in, dim_out, stride) if (dim_in != dim_out or stride != 1) else None


class X3DFeatures(_plan.HipTrunk):
    """X3D-S / X3D-M trunk up to res5: (N,3,T,H,W) -> (N,192,T,H/32,W/32)."""

    out_channels = 192

    def __init__(self, arch: str = "x3d_s"):
        super().__init__()
        self.arch = arch
        self.stem = _X3DStem(X3D_STEM_DIM)
        dim_in = X3D_STEM_DIM
        stages = []
        for dim_out, depth in X3D_STAGES:
            blocks = []
            for bi in range(depth):
                blocks.append(_X3DBlock(dim_in, dim_out, 2 if bi == 0 else 1, use_se=(bi % 2 == 0)))
                dim_in = dim_out
            stages.append(nn.Sequential(*blocks))
        self.stages = nn.Sequential(*stages)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def __repr__(self):
        return f"{self.arch}_features(stem={X3D_STEM_DIM}, stages={X3D_STAGES})"

    def build_plan(self, pb, x):
        x = pb.x3d_stem(x, self.stem.conv_xy, self.stem.conv_t, self.stem.bn)  # one T-marching launch (two if unsupported)
        for stage in self.stages:
            blocks = list(stage)
            pre = None  # this block's expand-conv output when the previous block's chained launch already made it
            edp = [False] * len(blocks)  # blocks that run as ONE launch (expand -> stencil -> project [-> next expand]): 7 x 7 planes, no squeeze-excite
            for i, blk in enumerate(blocks):
                sc = x
                if i == 0:  # (behind the stage's first block the shape is fixed: decide for the whole stage before any pairing)
                    sb = _plan._triple(blk.conv_b.stride, 1)
                    co = blk.conv_c.out_channels
                    xs = _plan.Act(x.N, x.T, (x.H - 1) // sb[1] + 1, (x.W - 1) // sb[2] + 1, co, _plan.round_up(co, 8), -1)
                    edp = [j >= 1 and b.se is None and b.shortcut is None and bool(pb.x3d_edp(xs, b.conv_a, b.bn_a, b.conv_b, b.bn_b, b.conv_c, b.bn_c, probe=True))
                           for j, b in enumerate(blocks)]
                if edp[i] and pre is None:
                    nb_ = blocks[i + 1] if i + 1 < len(blocks) else None
                    chain = nb_ is not None and nb_.shortcut is None and not edp[i + 1]
                    whole = pb.x3d_edp(x, blk.conv_a, blk.bn_a, blk.conv_b, blk.bn_b, blk.conv_c, blk.bn_c, nb_.conv_a if chain else None, nb_.bn_a if chain else None)
                    if whole is not None:
                        x, pre = whole
                        continue
                fuse_short = blk.shortcut is not None and pb.short_fusable(x, blk)  # the strided shortcut conv rides in the project conv's launch
                if blk.shortcut is not None and not fuse_short:
                    sc = pb.conv(x, blk.shortcut.conv, blk.shortcut.bn, act="none")
                act_b = "none" if blk.se is not None else "swish"  # no gate between BN and Swish: the stencil applies Swish
                gate = None
                # expand conv + stencil in one launch where the pair is covered (block width <= 48): the expanded activation never leaves LDS
                front = pb.expand_dw(x, blk.conv_a, blk.bn_a, blk.conv_b, blk.bn_b, act_b, pool=blk.se is not None) if pre is None else None
                if front is not None:
                    if blk.se is not None:
                        y, pooled = front
                        gate = pb.se_gate_or_prologue(y, pooled, blk.se.fc1, blk.se.fc2, consumer=(blk.conv_c, True))
                    else:
                        y = front
                e = None if front is not None else pre if pre is not None else pb.conv(x, blk.conv_a, blk.bn_a, act="relu")
                if front is not None:
                    pass
                elif blk.se is not None:
                    # stencil + gate in one launch where that pays; where the project conv can compute the gate in its own prologue, only
                    # the pool partial rows are produced here (gate = ("pooled", ...))
                    y, gate = pb.dwconv_se(e, blk.conv_b, blk.bn_b, blk.se.fc1, blk.se.fc2, consumer=(blk.conv_c, True))
                else:
                    y = pb.dwconv(e, blk.conv_b, blk.bn_b, act=act_b)
                # project conv; where the geometry allows, chained in ONE launch with the next block's expand conv
                nxt = blocks[i + 1] if i + 1 < len(blocks) else None
                pair = None
                if isinstance(gate, tuple):  # squeeze-excite gate in the project conv's prologue
                    if fuse_short:  # (not combined with the fused shortcut: different kernels)
                        sc = pb.conv(x, blk.shortcut.conv, blk.shortcut.bn, act="none")
                        fuse_short = False
                    if nxt is not None and nxt.shortcut is None and not edp[i + 1]:  # ... chained with the next block's expand conv where that is covered (a whole-block launch makes its own)
                        pair = pb.conv_pair(y, blk.conv_c, blk.bn_c, "relu", sc, nxt.conv_a, nxt.bn_a, "relu", in_swish=True,
                                            se=(gate[1], blk.se.fc1, blk.se.fc2))
                        if pair is not None:
                            x, pre = pair
                            continue
                    yc = pb.conv_se(y, blk.conv_c, blk.bn_c, "relu", sc, gate[1], blk.se.fc1, blk.se.fc2)
                    if yc is not None:
                        x, pre = yc, None
                        continue
                    gate = pb.se_gate(gate[1], blk.se.fc1, blk.se.fc2)
                if fuse_short:
                    fused = pb.conv_short(y, blk.conv_c, blk.bn_c, "relu", x, blk.shortcut.conv, blk.shortcut.bn, in_gate=gate,
                                          in_swish=blk.se is not None)
                    if fused is None:  # (short_fusable said yes on the same descriptors; kept for safety)
                        sc = pb.conv(x, blk.shortcut.conv, blk.shortcut.bn, act="none")
                    else:
                        x, pre = fused, None
                        continue
                if nxt is not None and nxt.shortcut is None and not edp[i + 1]:
                    pair = pb.conv_pair(y, blk.conv_c, blk.bn_c, "relu", sc, nxt.conv_a, nxt.bn_a, "relu",
                                        in_gate=gate, in_swish=blk.se is not None)
                if pair is not None:
                    x, pre = pair
                else:
                    x = pb.conv(y, blk.conv_c, blk.bn_c, act="relu", residual=sc, in_gate=gate, in_swish=blk.se is not None)
                    pre = None
        return x


    def build_train(self, tb, x):
        """Training pass (batch-statistics norm, every unit's raw conv output and activation kept for the backward)."""
        e = tb.unit(x, self.stem.conv_xy, None, "none", kind="first")
        x = tb.unit(e, self.stem.conv_t, self.stem.bn, "relu", kind="dw")
        for stage in self.stages:
            for blk in stage:
                sc = x if blk.shortcut is None else tb.unit(x, blk.shortcut.conv, blk.shortcut.bn, "none")
                a = tb.unit(x, blk.conv_a, blk.bn_a, "relu")
                b = tb.unit(a, blk.conv_b, blk.bn_b, "swish", kind="dw", se=blk.se)
                x = tb.unit(b, blk.conv_c, blk.bn_c, "relu", residual=sc)
        return x


def x3d_s(pretrained: bool = False, **kwargs) -> X3DFeatures:
    if pretrained:
        raise RuntimeError("x3d_*: the reference ships no X3D trunk and no weight file for one (resnet_features.py:8-16); use pretrained=False")
    return X3DFeatures("x3d_s")


def x3d_m(pretrained: bool = False, **kwargs) -> X3DFeatures:
    if pretrained:
        raise RuntimeError("x3d_*: the reference ships no X3D trunk and no weight file for one (resnet_features.py:8-16); use pretrained=False")
    return X3DFeatures("x3d_m")


# registry: reference src/models/ProtoPNet.py:35-54 lists 18 trunks; the shipped configs select only
# resnet18 and resnet2p1d_18.  x3d_* are the BASELINE.json additions.
base_architecture_to_features = {
    "resnet2p1d_18": resnet2p1d_18,
    "resnet18": resnet18_features,
    "x3d_s": x3d_s,
    "x3d_m": x3d_m,
}
