"""``TransformLoss`` of the reference (src/loss/loss.py:257-320) on the HIP path.

The reference warps the input clip and the occurrence maps with ``torchvision.transforms.functional.affine`` (rotation
in [-20, 20] degrees, scale in [0.6, 1.5], bilinear, fill 0), runs ``model.compute_occurence_map`` on the warped clip and takes
the L1 distance between the two sets of maps.  Here the warp is ``pasn_affine_warp_fwd`` / ``_bwd`` (csrc/warp.hip) under a small
``autograd.Function``; the second trunk pass is the compiled training pass of ``compute_occurence_map``; the L1 reduction over
the (N, P, T', H', W') maps is a plain torch op on a tiny tensor.  Same constructor arguments and ``compute`` signature as the
reference class, so an agent swaps the import only.  No torchvision is needed (it is absent from this image)."""
from __future__ import annotations

import random

import torch

from . import _lib


class _AffineWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, angle, scale):
        if not x.is_cuda:
            raise RuntimeError("protoasnet_amd.losses run on the GPU only; there is no CPU fallback")
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        h, w = x.shape[-2], x.shape[-1]
        planes = x.numel() // (h * w)
        y = torch.empty_like(x)
        _lib.check(_lib.lib().pasn_affine_warp_fwd(x.data_ptr(), y.data_ptr(), planes, h, w, float(angle), float(scale), _lib.dtype_code(x.dtype),
                                                   _lib.current_stream()))
        ctx.geom = (planes, h, w, float(angle), float(scale), x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        planes, h, w, angle, scale, dtype = ctx.geom
        dyf = dy.contiguous().float()
        dx = torch.zeros_like(dyf)
        _lib.check(_lib.lib().pasn_affine_warp_bwd(dyf.data_ptr(), dx.data_ptr(), planes, h, w, angle, scale, _lib.current_stream()))
        return dx.to(dtype), None, None


def affine_warp(x: torch.Tensor, angle: float, scale: float) -> torch.Tensor:
    """Every trailing (H, W) plane of ``x`` rotated by ``angle`` degrees about its centre and scaled by ``scale`` (torchvision
    ``affine`` with translate = (0, 0), shear = 0, bilinear interpolation, fill = 0)."""
    return _AffineWarp.apply(x, angle, scale)


def get_affine_config() -> dict:
    """The reference's sampler (loss.py:257-269): the keys that vary."""
    return {"angle": random.uniform(-20, 20), "scale": random.uniform(0.6, 1.5)}


class TransformLoss(object):
    """reference src/loss/loss.py:272-320 -- same arguments, same ``compute(x, occurrence_map, model)``."""

    def __init__(self, loss_weight=1e-4, reduction="sum"):
        self.loss_weight = loss_weight
        self.reduction = reduction

    def compute(self, x, occurrence_map, model, config=None):
        if self.loss_weight == 0:
            return torch.tensor(0, device=x.device)
        cfg = config or get_affine_config()
        transformed_x = affine_warp(x, cfg["angle"], cfg["scale"])  # per-frame 2-D warp of (N,3,[T,]H,W)
        occ_t = model.compute_occurence_map(transformed_x).squeeze(2)  # (N, P, [T',] H', W')
        warped = affine_warp(occurrence_map.squeeze(2), cfg["angle"], cfg["scale"])
        loss = (occ_t - warped).abs().sum()
        if self.reduction == "mean":
            loss = loss / (occ_t.shape[0] * occ_t.shape[1])
        return self.loss_weight * loss
