"""Oracle prototype layers (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Two heads exist in the reference:

* head A -- ``PPNet``: squared-L2 distance map, global min, log activation;
* head B -- ``XProtoNet`` / ``Video_XProtoNet`` ("ProtoASNet"): occurrence-map
  weighted pooling followed by cosine similarity.

The op order of the reference is kept literally (including the redundant
ones-convolution and the materialised broadcast product) because the oracle also
serves as the reported CPU baseline.
"""
from __future__ import annotations

from typing import Mapping, Tuple

import torch
import torch.nn.functional as F


def _conv1x1(x: torch.Tensor, w: torch.Tensor, b=None) -> torch.Tensor:
    return F.conv3d(x, w, b) if x.dim() == 5 else F.conv2d(x, w, b)


def _chain_indices(sd: Mapping[str, torch.Tensor], name: str):
    idx = sorted(int(k.split(".")[1]) for k in sd if k.startswith(name + ".") and k.endswith(".weight"))
    return idx


def add_on_layers(sd, x: torch.Tensor, final_sigmoid: bool) -> torch.Tensor:
    """1x1(x1) conv chain with ReLU between convs.

    PPNet keeps a trailing Sigmoid (src/models/ProtoPNet.py:91-130); XProtoNet strips it
    (src/models/XProtoNet.py:17); Video_XProtoNet never had one (src/models/Video_XProtoNet.py:27-39).
    """
    idx = _chain_indices(sd, "add_on_layers")
    for j, i in enumerate(idx):
        x = _conv1x1(x, sd[f"add_on_layers.{i}.weight"], sd.get(f"add_on_layers.{i}.bias"))
        if j + 1 < len(idx):
            x = F.relu(x)
        elif final_sigmoid:
            x = torch.sigmoid(x)
    return x


def occurrence_logits(sd, x: torch.Tensor) -> torch.Tensor:
    """``self.occurrence_module(x)``: conv-ReLU-conv-ReLU-conv(no bias), (N, P, [T,] H, W).

    Reference: src/models/Video_XProtoNet.py:42-62; src/models/XProtoNet.py:21-41.
    """
    idx = _chain_indices(sd, "occurrence_module")
    for j, i in enumerate(idx):
        x = _conv1x1(x, sd[f"occurrence_module.{i}.weight"], sd.get(f"occurrence_module.{i}.bias"))
        if j + 1 < len(idx):
            x = F.relu(x)
    return x


def occurrence_map_abs(sd, x: torch.Tensor) -> torch.Tensor:
    """``get_occurence_map_absolute_val``: the occurrence module, abs, unsqueeze(2).

    Reference: src/models/Video_XProtoNet.py:106-109; src/models/XProtoNet.py:82-85.
    """
    return torch.abs(occurrence_logits(sd, x)).unsqueeze(2)


def occurrence_map_softmaxed(sd, x: torch.Tensor) -> torch.Tensor:
    """``get_occurence_map_softmaxed`` (src/models/XProtoNet.py:75-80): softmax over the flattened positions, unsqueeze(2)."""
    om = occurrence_logits(sd, x)
    n, p = om.shape[:2]
    return torch.softmax(om.reshape(n, p, -1), dim=-1).reshape(om.shape).unsqueeze(2)


# ------------------------------------------------------------------ head A (PPNet)
def l2_convolution(x: torch.Tensor, prototype_vectors: torch.Tensor, ones: torch.Tensor) -> torch.Tensor:
    """``PPNet._l2_convolution`` -- src/models/ProtoPNet.py:189-207."""
    x2_patch_sum = F.conv2d(x**2, ones)
    p2 = torch.sum(prototype_vectors**2, dim=(1, 2, 3)).view(-1, 1, 1)
    xp = F.conv2d(x, prototype_vectors)
    return F.relu(x2_patch_sum + (-2 * xp + p2))


def distance_2_similarity(d: torch.Tensor, activation="log", epsilon: float = 1e-4) -> torch.Tensor:
    """src/models/ProtoPNet.py:217-223 (epsilon = 1e-4 at :74)."""
    if activation == "log":
        return torch.log((d + 1) / (d + epsilon))
    if activation == "linear":
        return -d
    return activation(d)


def ppnet_head(sd, conv_features: torch.Tensor, activation="log", epsilon: float = 1e-4):
    """Distance map -> global min -> activation -> last layer.  src/models/ProtoPNet.py:229-243."""
    distances = l2_convolution(conv_features, sd["prototype_vectors"], sd["ones"])
    min_d = -F.max_pool2d(-distances, kernel_size=(distances.size(2), distances.size(3)))
    min_d = min_d.view(-1, sd["prototype_vectors"].shape[0])
    act = distance_2_similarity(min_d, activation, epsilon)
    logits = F.linear(act, sd["last_layer.weight"])
    return {"logits": logits, "min_distances": min_d, "distances": distances}


# ------------------------------------------------------------------ head B (XProtoNet / Video)
def cosine_similarity_dim2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``nn.CosineSimilarity(dim=2)`` (eps=1e-8) as torch 2.10 evaluates it."""
    return F.cosine_similarity(a, b, dim=2, eps=1e-8)


def xproto_head(sd, x: torch.Tensor, contract: bool = False):
    """Everything after the trunk in ``XProtoNet.forward`` / ``Video_XProtoNet.forward``.

    Reference: src/models/XProtoNet.py:51-67 and src/models/Video_XProtoNet.py:82-98
    (``push_forward`` :111-130 returns the same tensors in another order plus ``1 - similarity``).
    ``contract=True`` replaces the materialised (N,P,D,...) broadcast product by the
    mathematically identical contraction; used only where the product would not fit
    in memory, never for the CPU baseline.
    """
    feature_map = add_on_layers(sd, x, final_sigmoid=False).unsqueeze(1)  # (N,1,D,[T],H,W)
    occurrence_map = occurrence_map_abs(sd, x)  # (N,P,1,[T],H,W)
    if contract:
        n, p = occurrence_map.shape[:2]
        d = feature_map.shape[2]
        features_extracted = torch.einsum(
            "nps,nds->npd", occurrence_map.reshape(n, p, -1), feature_map.reshape(n, d, -1)
        )
    else:
        prod = occurrence_map * feature_map
        features_extracted = prod.sum(dim=3).sum(dim=3)
        if x.dim() == 5:
            features_extracted = features_extracted.sum(dim=3)
    protos = sd["prototype_vectors"].squeeze().unsqueeze(0)
    similarity = (cosine_similarity_dim2(features_extracted, protos) + 1) / 2.0
    logits = F.linear(similarity, sd["last_layer.weight"])
    return {
        "logits": logits,
        "similarity": similarity,
        "occurrence_map": occurrence_map,
        "features_extracted": features_extracted,
    }


# ------------------------------------------------------------------ constructor semantics
def prototype_class_identity(num_prototypes: int, num_classes: int) -> torch.Tensor:
    """One-hot (P, K), block layout.  src/models/ProtoPNet.py:326-340."""
    assert num_prototypes % num_classes == 0
    ident = torch.zeros(num_prototypes, num_classes)
    per = num_prototypes // num_classes
    for j in range(num_prototypes):
        ident[j, j // per] = 1
    return ident


def last_layer_init(identity: torch.Tensor, incorrect_strength: float) -> torch.Tensor:
    """src/models/ProtoPNet.py:299-311."""
    pos = torch.t(identity)
    return 1 * pos + incorrect_strength * (1 - pos)
