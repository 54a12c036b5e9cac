"""Oracle whole-model passes (TEST INFRASTRUCTURE -- see oracle/__init__.py).

``forward`` / ``push_forward`` / ``compute_occurence_map`` of the three reference
models, composed from ``oracle.backbones`` and ``oracle.heads`` over a
``state_dict`` with the reference's key names.
"""
from __future__ import annotations

from typing import Mapping

import torch

from . import backbones, heads


@torch.no_grad()
def ppnet_forward(sd: Mapping[str, torch.Tensor], x, arch="resnet18", activation="log", epsilon=1e-4):
    """``PPNet.forward`` (src/models/ProtoPNet.py:225-243) and ``push_forward`` (:245-249) products."""
    feat = backbones.trunk(arch, sd, "features.", x)
    conv_features = heads.add_on_layers(sd, feat, final_sigmoid=True)  # conv_features(), :144-150
    out = heads.ppnet_head(sd, conv_features, activation, epsilon)
    out["conv_features"] = conv_features
    out["backbone_features"] = feat
    return out


@torch.no_grad()
def xprotonet_forward(sd, x, arch="resnet18", last_layer_num=-3, contract=False):
    """``XProtoNet.forward`` (src/models/XProtoNet.py:51-67) / ``Video_XProtoNet.forward``
    (src/models/Video_XProtoNet.py:82-98); ``push_forward`` reorders the same products and adds
    ``1 - similarity`` (XProtoNet.py:87-106, Video_XProtoNet.py:111-130).
    """
    feat = backbones.trunk(arch, sd, "cnn_backbone.", x, last_layer_num)
    out = heads.xproto_head(sd, feat, contract=contract)
    out["backbone_features"] = feat
    out["proto_dist"] = 1 - out["similarity"]
    return out


@torch.no_grad()
def compute_occurence_map(sd, x, arch="resnet18", last_layer_num=-3):
    """src/models/XProtoNet.py:69-73, src/models/Video_XProtoNet.py:100-104."""
    feat = backbones.trunk(arch, sd, "cnn_backbone.", x, last_layer_num)
    return heads.occurrence_map_abs(sd, feat)


def xprotonet_train_forward(sd, x, arch="x3d_s", last_layer_num=-3, occurrence_only=False):
    """The same passes in TRAIN mode with autograd enabled: what ``loss.backward()`` differentiates in the reference's
    training loop (src/agents/Video_XProtoNet_e2e.py:118-141; ``compute_occurence_map`` with gradients at src/loss/loss.py:302).
    Running statistics in ``sd`` are updated in place."""
    with backbones.train_mode():
        feat = backbones.trunk(arch, sd, "cnn_backbone.", x, last_layer_num)
    if occurrence_only:
        return {"occurrence_map": heads.occurrence_map_abs(sd, feat)}
    return heads.xproto_head(sd, feat, contract=True)


def ppnet_train_forward(sd, x, arch="resnet18", activation="log", epsilon=1e-4):
    """``PPNet.forward`` (src/models/ProtoPNet.py:225-243) in TRAIN mode with autograd enabled (ProtoPNet_Base.py trains through it)."""
    with backbones.train_mode():
        feat = backbones.trunk(arch, sd, "features.", x)
    conv_features = heads.add_on_layers(sd, feat, final_sigmoid=True)
    return heads.ppnet_head(sd, conv_features, activation, epsilon)
