"""Oracle feature trunks (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Functional fp32 restatements over a ``state_dict``; ``prefix`` is the key prefix
of the trunk inside the model (``"features."`` for PPNet, ``"cnn_backbone."``
for XProtoNet / Video_XProtoNet).  Norm layers run in eval mode (running
statistics), which is what the reference uses for the ``forward`` clips/s metric
(``model.eval()`` at ``src/agents/Video_XProtoNet_e2e.py:41``) and for push
(``src/utils/push_abs_revision.py:210``).
"""
from __future__ import annotations

import math
from typing import Mapping

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm{2,3}d default, never overridden by the reference


BN_MOMENTUM = 0.1  # torch default
_TRAIN = [False]


class train_mode:
    """``with train_mode():`` -- norm layers use batch statistics and update the running estimates in ``sd`` in place, as
    ``model.train()`` does in the reference's training epochs (src/agents/Video_XProtoNet_e2e.py:118)."""

    def __enter__(self):
        self._old = _TRAIN[0]
        _TRAIN[0] = True

    def __exit__(self, *exc):
        _TRAIN[0] = self._old


def _bn(sd: Mapping[str, torch.Tensor], p: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(
        x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], _TRAIN[0],
        BN_MOMENTUM if _TRAIN[0] else 0.0, BN_EPS
    )


# --------------------------------------------------------------------------------------
# 2-D ResNet-18 feature trunk -- PINNED by golden vectors (reference code runs here).
# --------------------------------------------------------------------------------------
def resnet18_features(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """``ResNet_features.forward`` with ``BasicBlock`` x [2,2,2,2].

    Reference: src/models/resnet_features.py:202-213 (trunk), :49-66 (BasicBlock.forward),
    :139-142 (7x7 s2 conv, 3x3 s2 max-pool), :177-200 (_make_layer: 1x1 strided downsample + BN
    on the first block of layer2-4).
    """
    x = F.conv2d(x, sd[prefix + "conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(sd, prefix + "bn1", x))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for b in range(2):
            p = f"{prefix}layer{li}.{b}."
            s = stride if b == 0 else 1
            identity = x
            out = F.conv2d(x, sd[p + "conv1.weight"], stride=s, padding=1)
            out = F.relu(_bn(sd, p + "bn1", out))
            out = F.conv2d(out, sd[p + "conv2.weight"], stride=1, padding=1)
            out = _bn(sd, p + "bn2", out)
            if (p + "downsample.0.weight") in sd:
                identity = F.conv2d(x, sd[p + "downsample.0.weight"], stride=s)
                identity = _bn(sd, p + "downsample.1", identity)
            x = F.relu(out + identity)
    return x


def resnet18_conv_info():
    """(kernel_sizes, strides, paddings) as accumulated by the reference trunk.

    Reference: src/models/resnet_features.py:144-146 (stem + max-pool), :68-73
    (BasicBlock.block_conv_info), :188-198 (per-block accumulation).
    """
    ks, st, pd = [7, 3], [2, 2], [3, 1]
    for stride in (1, 2, 2, 2):
        for b in range(2):
            ks += [3, 3]
            st += [stride if b == 0 else 1, 1]
            pd += [1, 1]
    return ks, st, pd


# --------------------------------------------------------------------------------------
# R(2+1)D-18 trunk -- PARITY UNPINNED (third-party torchvision graph, absent here).
# --------------------------------------------------------------------------------------
def r2plus1d_midplanes(inplanes: int, planes: int) -> int:
    """torchvision 0.14.1 ``video/resnet.py`` BasicBlock: one midplanes value per block."""
    return (inplanes * planes * 3 * 3 * 3) // (inplanes * 3 * 3 + 3 * planes)


def _conv2plus1d(sd, p: str, x, stride: int):
    """torchvision ``Conv2Plus1D``: (1,3,3) conv -> BN -> ReLU -> (3,1,1) conv."""
    x = F.conv3d(x, sd[p + "0.weight"], stride=(1, stride, stride), padding=(0, 1, 1))
    x = F.relu(_bn(sd, p + "1", x))
    x = F.conv3d(x, sd[p + "3.weight"], stride=(stride, 1, 1), padding=(1, 0, 0))
    return x


def r2plus1d_18_trunk(sd, prefix: str, x: torch.Tensor, last_layer_num: int = -3) -> torch.Tensor:
    """``resnet2p1d_18.forward``: ``nn.Sequential(*children(r2plus1d_18)[:last_layer_num])``.

    Reference call site: src/models/resnet_features.py:316-327; children order is
    [stem, layer1, layer2, layer3, layer4, avgpool, fc], so -3 keeps stem+layer1-3
    (256 channels, T/4, H/8, W/8 -- :311-313) and -2 also keeps layer4.
    Graph restated from torchvision 0.14.1 ``models/video/resnet.py``
    (R2Plus1dStem, Conv2Plus1D, BasicBlock, VideoResNet._make_layer).
    """
    n_children = 7 + last_layer_num
    assert 1 <= n_children <= 5, "only stem..layer4 are convolutional children"
    p = prefix + "backbone.0."
    x = F.conv3d(x, sd[p + "0.weight"], stride=(1, 2, 2), padding=(0, 3, 3))
    x = F.relu(_bn(sd, p + "1", x))
    x = F.conv3d(x, sd[p + "3.weight"], stride=(1, 1, 1), padding=(1, 0, 0))
    x = F.relu(_bn(sd, p + "4", x))
    for li in range(1, n_children):
        stride = 1 if li == 1 else 2
        for b in range(2):
            q = f"{prefix}backbone.{li}.{b}."
            s = stride if b == 0 else 1
            identity = x
            out = _conv2plus1d(sd, q + "conv1.0.", x, s)
            out = F.relu(_bn(sd, q + "conv1.1", out))
            out = _conv2plus1d(sd, q + "conv2.0.", out, 1)
            out = _bn(sd, q + "conv2.1", out)
            if (q + "downsample.0.weight") in sd:
                identity = F.conv3d(x, sd[q + "downsample.0.weight"], stride=(s, s, s))
                identity = _bn(sd, q + "downsample.1", identity)
            x = F.relu(out + identity)
    return x


# --------------------------------------------------------------------------------------
# X3D trunk -- PARITY UNPINNED (not in the reference; named by BASELINE.json).
# --------------------------------------------------------------------------------------
X3D_STEM_DIM = 24
X3D_STAGES = ((24, 3), (48, 5), (96, 11), (192, 7))  # (dim_out, depth); X3D-S and X3D-M share it
X3D_BOTTLENECK = 2.25
X3D_SE_RATIO = 0.0625


def x3d_round_width(width: float, multiplier: float, min_width: int = 8, divisor: int = 8) -> int:
    """Channel rounding of the X3D paper's reference implementation (pytorchvideo ``round_width``)."""
    width *= multiplier
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


def x3d_trunk(sd, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """X3D-S/M feature trunk up to res5 (192 channels, T, H/32, W/32).

    Definition (Feichtenhofer, "X3D", CVPR 2020, table 3; layer recipe of
    pytorchvideo ``create_x3d``): stem = (1,3,3) s(1,2,2) conv 3->24, depthwise
    (5,1,1) conv, BN, ReLU; four stages of bottleneck blocks
    [1x1x1 expand x2.25 -> BN -> ReLU -> depthwise 3x3x3 (stride (1,2,2) on the first
    block of a stage) -> BN -> SE (ratio 1/16, even block indices only) -> Swish ->
    1x1x1 project -> BN] + shortcut (1x1x1 strided conv + BN on the first block),
    ReLU after the sum.
    """
    x = F.conv3d(x, sd[prefix + "stem.conv_xy.weight"], stride=(1, 2, 2), padding=(0, 1, 1))
    x = F.conv3d(x, sd[prefix + "stem.conv_t.weight"], padding=(2, 0, 0), groups=x.shape[1])
    x = F.relu(_bn(sd, prefix + "stem.bn", x))
    for si, (dim_out, depth) in enumerate(X3D_STAGES):
        for bi in range(depth):
            p = f"{prefix}stages.{si}.{bi}."
            s = 2 if bi == 0 else 1
            sc = x
            if (p + "shortcut.conv.weight") in sd:
                sc = F.conv3d(x, sd[p + "shortcut.conv.weight"], stride=(1, s, s))
                sc = _bn(sd, p + "shortcut.bn", sc)
            y = F.conv3d(x, sd[p + "conv_a.weight"])
            y = F.relu(_bn(sd, p + "bn_a", y))
            y = F.conv3d(y, sd[p + "conv_b.weight"], stride=(1, s, s), padding=1, groups=y.shape[1])
            y = _bn(sd, p + "bn_b", y)
            if (p + "se.fc1.weight") in sd:
                g = y.mean(dim=(2, 3, 4), keepdim=True)
                g = F.relu(F.conv3d(g, sd[p + "se.fc1.weight"], sd[p + "se.fc1.bias"]))
                g = torch.sigmoid(F.conv3d(g, sd[p + "se.fc2.weight"], sd[p + "se.fc2.bias"]))
                y = y * g
            y = y * torch.sigmoid(y)  # Swish
            y = F.conv3d(y, sd[p + "conv_c.weight"])
            y = _bn(sd, p + "bn_c", y)
            x = F.relu(sc + y)
    return x


def trunk_out_shape(arch: str, in_shape, last_layer_num: int = -3):
    """Feature-map shape (C, T', H', W') or (C, H', W') a trunk yields for ``in_shape`` = (3, [T,] H, W)."""

    def half(v, k, s, p):
        return (v + 2 * p - k) // s + 1

    if arch == "resnet18":
        _, h, w = in_shape
        for k, s, p in ((7, 2, 3), (3, 2, 1), (3, 2, 1), (3, 2, 1), (3, 2, 1)):
            h, w = half(h, k, s, p), half(w, k, s, p)
        return (512, h, w)
    if arch == "resnet2p1d_18":
        _, t, h, w = in_shape
        h, w = half(h, 7, 2, 3), half(w, 7, 2, 3)
        c = 64
        for li in range(2, 7 + last_layer_num):
            t, h, w = half(t, 3, 2, 1), half(h, 3, 2, 1), half(w, 3, 2, 1)
            c *= 2
        return (c, t, h, w)
    if arch in ("x3d_s", "x3d_m"):
        _, t, h, w = in_shape
        for _ in range(5):
            h, w = half(h, 3, 2, 1), half(w, 3, 2, 1)
        return (192, t, h, w)
    raise ValueError(arch)


def trunk(arch: str, sd, prefix: str, x: torch.Tensor, last_layer_num: int = -3) -> torch.Tensor:
    if arch == "resnet18":
        return resnet18_features(sd, prefix, x)
    if arch == "resnet2p1d_18":
        return r2plus1d_18_trunk(sd, prefix, x, last_layer_num)
    if arch in ("x3d_s", "x3d_m"):
        return x3d_trunk(sd, prefix, x)
    raise ValueError(f"oracle has no trunk for base_architecture={arch!r}")
