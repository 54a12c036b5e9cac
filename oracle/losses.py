"""Oracle of ``TransformLoss`` (TEST INFRASTRUCTURE -- see oracle/__init__.py).  PARITY UNPINNED for the warp: the reference calls
``torchvision.transforms.functional.affine`` (src/loss/loss.py:4,257-320) and torchvision is not installed in this image, so there
are no reference outputs to pin against; this restates the torchvision 0.14 tensor path (functional_tensor.py:
``_get_inverse_affine_matrix`` with centre (0, 0), translate (0, 0), shear 0 -> ``_gen_affine_grid`` ->
``grid_sample(mode="bilinear", padding_mode="zeros", align_corners=False)`` with a ones channel appended for ``fill``) with the
same torch ops it uses (0.14.1 = the version of the reference's docker image, README.md:44)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def affine(img: torch.Tensor, angle: float, scale: float) -> torch.Tensor:
    """``torchvision.transforms.functional.affine(img, angle, (0,0), scale, 0.0, BILINEAR, fill=0)`` for (M, C, H, W) tensors."""
    m, c, h, w = img.shape
    rot = math.radians(angle)
    # _get_inverse_affine_matrix(center=[0,0], angle, translate=[0,0], scale, shear=[0,0]): M^-1 = R(-angle)^-1 / scale
    theta = torch.tensor([[math.cos(rot), math.sin(rot), 0.0], [-math.sin(rot), math.cos(rot), 0.0]], dtype=img.dtype) / scale
    # _gen_affine_grid: pixel-centre base grid, rescaled to grid_sample's normalised coordinates
    d = 0.5
    base = torch.empty(1, h, w, 3, dtype=img.dtype)
    base[..., 0] = torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, w).view(1, 1, w)
    base[..., 1] = torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, h).view(1, h, 1)
    base[..., 2] = 1.0
    rescaled = theta.t() / torch.tensor([0.5 * w, 0.5 * h], dtype=img.dtype)
    grid = base.view(1, h * w, 3).matmul(rescaled).view(1, h, w, 2).expand(m, h, w, 2)
    # _apply_grid_transform with fill: ones channel, sample, blend
    x = torch.cat((img, torch.ones(m, 1, h, w, dtype=img.dtype)), dim=1)
    x = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    mask = x[:, -1:, :, :]
    return x[:, :-1, :, :] * mask  # + (1 - mask) * fill with fill = 0


def transform_loss(x, occurrence_map, occurrence_map_of, angle, scale, loss_weight=1e-4, reduction="sum"):
    """loss.py:283-320 with the sampled configuration passed in; ``occurrence_map_of(x)`` plays ``model.compute_occurence_map``."""
    video = x.dim() == 5
    if video:
        n, dch, t, h, w = x.shape
        xt = affine(x.permute(0, 2, 1, 3, 4).reshape(-1, dch, h, w), angle, scale).reshape(n, t, dch, h, w).permute(0, 2, 1, 3, 4)
    else:
        xt = affine(x, angle, scale)
    occ_t = occurrence_map_of(xt).squeeze(2)
    occ = occurrence_map.squeeze(2)
    if video:
        n, p, t, h, w = occ.shape
        occ_w = affine(occ.permute(0, 2, 1, 3, 4).reshape(-1, p, h, w), angle, scale).reshape(n, t, p, h, w).permute(0, 2, 1, 3, 4)
    else:
        occ_w = affine(occ, angle, scale)
    loss = F.l1_loss(occ_t, occ_w, reduction="sum")
    if reduction == "mean":
        loss = loss / (occ_t.shape[0] * occ_t.shape[1])
    return loss_weight * loss


# --------------------------------------------------------------------------------------------------------------------------
# The scalar loss terms of the reference's training step (src/loss/loss.py), restated as plain functions for oracle/trainer.py.
# PINNED: tests/test_cpu_losses.py::test_oracle_loss_terms_match_reference_golden compares value and gradient of every one with
# tests/golden/g6_losses.npz, which tests/golden/make_golden_losses.py produced by RUNNING the reference's classes here.
# --------------------------------------------------------------------------------------------------------------------------
def _zero(t):
    return torch.tensor(0, device=t.device)  # loss.py:29,52,82,... the reference's zero-weight return value


def ce_loss(logits, target, loss_weight=1, reduction="mean"):
    """``CeLoss.compute`` (loss.py:23-34)."""
    if loss_weight == 0:
        return _zero(target)
    return loss_weight * F.cross_entropy(logits, target, reduction=reduction)


def ce_loss_abstain(logits, target, loss_weight=1, ab_weight=0.3, reduction="sum", ab_logitpath="joined"):
    """``CeLossAbstain.compute`` (loss.py:323-371): the K+1-th logit is a learned abstention probability."""
    if loss_weight == 0:
        return _zero(target)
    k = logits.shape[1] - 1
    if ab_logitpath == "joined":
        abstention = logits.softmax(dim=1)[:, k:k + 1]
    else:
        abstention = logits.sigmoid()[:, k:k + 1]
    pred = logits[:, :k].softmax(dim=1)
    virtual = (1 - abstention) * pred + abstention * F.one_hot(target, num_classes=k)
    loss_pred = F.nll_loss(torch.log(virtual), target, reduction=reduction)
    loss_abs = -torch.log(1 - abstention).squeeze()
    if reduction == "mean":
        loss_abs = loss_abs.mean()
    elif reduction == "sum":
        loss_abs = loss_abs.sum()
    return loss_weight * (loss_pred + ab_weight * loss_abs)


def _class_max(similarities, num_classes):
    return similarities.reshape(similarities.shape[0], num_classes, -1).max(dim=2)[0]  # (N, classes): loss.py:127-129


def _reduce(per_class, reduction):
    return per_class.mean(dim=0).sum() if reduction == "mean" else per_class.sum()  # loss.py:134-137


def cluster_roi_feat(similarities, target, loss_weight, num_classes=4, reduction="sum"):
    """``ClusterRoiFeat.compute`` (loss.py:98-138)."""
    if loss_weight == 0:
        return _zero(target)
    positives = _class_max(similarities, num_classes) * F.one_hot(target, num_classes=num_classes)
    return loss_weight * _reduce(-1 * positives, reduction)


def separation_roi_feat(similarities, target, loss_weight, num_classes=4, reduction="sum", abstain_class=True):
    """``SeparationRoiFeat.compute`` (loss.py:141-183): the last class's prototypes are never penalised under ``abstain_class``."""
    if loss_weight == 0:
        return _zero(target)
    one_hot = F.one_hot(target, num_classes=num_classes)
    if abstain_class:
        one_hot[:, -1] = 1
    return loss_weight * _reduce(_class_max(similarities, num_classes) * (1 - one_hot), reduction)


def orthogonality(prototype_vectors, loss_weight, num_classes=4, mode="per_class"):
    """``OrthogonalityLoss.compute`` (loss.py:186-229)."""
    if loss_weight == 0:
        return _zero(prototype_vectors)
    if mode == "per_class":
        p = prototype_vectors.reshape(num_classes, -1, prototype_vectors.shape[1])
        sim = F.cosine_similarity(p.unsqueeze(1), p.unsqueeze(2), dim=3)
    else:
        p = prototype_vectors.squeeze()
        sim = F.cosine_similarity(p.unsqueeze(1), p.unsqueeze(0), dim=2)
    return loss_weight * torch.triu(sim, diagonal=1).sum()


def l_norm(tensor, dim=None, mask=None, p=1, loss_weight=1e-4, reduction="sum"):
    """``L_norm.compute`` (loss.py:232-254)."""
    if loss_weight == 0:
        return _zero(tensor)
    t = tensor if mask is None else mask.to(tensor.device) * tensor
    loss = t.norm(p=p, dim=dim)
    if reduction == "mean":
        loss = loss.mean(dim=0).sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss_weight * loss
