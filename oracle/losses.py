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
