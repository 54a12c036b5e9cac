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
        return self.compute_from_maps(occurrence_map, model.compute_occurence_map(transformed_x), cfg)

    def paired_forward(self, x, model, config=None):
        """The model's forward AND this term from ONE trunk pass over [x, warp(x)] (``model.forward_pair``): returns
        ``((logits, similarity, occurrence_map), loss_term)``.  The affine parameters are drawn here, i.e. BEFORE the forward where
        ``compute`` draws them after it; nothing in between consumes random numbers, so a seeded run sees the same transforms."""
        if self.loss_weight == 0 or not hasattr(model, "forward_pair"):
            out = model(x)
            return out, (torch.zeros((), device=x.device) if self.loss_weight == 0 else self.compute(x, out[2], model, config))
        cfg = config or get_affine_config()
        out, occ_t = model.forward_pair(x, affine_warp(x, cfg["angle"], cfg["scale"]))
        return out, self.compute_from_maps(out[2], occ_t, cfg)

    def compute_from_maps(self, occurrence_map, occurrence_map_transformed, config):
        """loss.py:302-320 once both sets of maps exist: ``occurrence_map_transformed`` = the model's maps of the warped clip (the
        caller may have produced them in the same pass as the originals: eval mode uses running statistics, so a 2N-clip batch of
        [clips, warped clips] gives each clip exactly what two N-clip passes give)."""
        occ_t = occurrence_map_transformed.squeeze(2)  # (N, P, [T',] H', W')
        warped = affine_warp(occurrence_map.squeeze(2), config["angle"], config["scale"])
        loss = (occ_t - warped).abs().sum()
        if self.reduction == "mean":
            loss = loss / (occ_t.shape[0] * occ_t.shape[1])
        return self.loss_weight * loss


# --------------------------------------------------------------------------------------------------------------------------
# The rest of the reference's loss stack (src/loss/loss.py): small reductions over (N, K) / (N, P) / (P, D) tensors, i.e. host
# plumbing in plain torch -- provided so a training step needs nothing from the reference's loss module (whose import drags in
# torchvision).  Same class names, constructor arguments and ``compute`` signatures; pinned against the reference's outputs and
# gradients by tests/golden/g6_losses.npz.
# --------------------------------------------------------------------------------------------------------------------------
def _zero(t):
    return torch.tensor(0, device=t.device)


def _per_class_extreme(scores, num_classes, largest):
    """(N, P) scores -> (N, classes): best prototype score of every class (prototypes are laid out class-major)."""
    grouped = scores.reshape(scores.shape[0], num_classes, -1)
    return grouped.max(dim=2)[0] if largest else grouped.min(dim=2)[0]


def _batch_reduce(per_class, reduction):
    """(N, classes) -> scalar: 'sum' over everything, 'mean' = batch mean then class sum (the reference's convention)."""
    return per_class.mean(dim=0).sum() if reduction == "mean" else per_class.sum()


class CeLoss(object):
    """loss.py:23-34"""

    def __init__(self, loss_weight=1, reduction="mean"):
        self.loss_weight, self.reduction = loss_weight, reduction

    def compute(self, logits, target):
        if self.loss_weight == 0:
            return _zero(target)
        return self.loss_weight * torch.nn.functional.cross_entropy(logits, target, reduction=self.reduction)


class ClusterPatch(object):
    """loss.py:37-65 -- ProtoPNet cluster cost on min_distances (N, P)."""

    def __init__(self, loss_weight, num_classes=4, reduction="mean"):
        self.loss_weight, self.num_classes, self.reduction = loss_weight, num_classes, reduction

    def compute(self, min_distances, target):
        if self.loss_weight == 0:
            return _zero(target)
        own = torch.nn.functional.one_hot(target, num_classes=self.num_classes)
        return self.loss_weight * _batch_reduce(_per_class_extreme(min_distances, self.num_classes, largest=False) * own, self.reduction)


class SeparationPatch(object):
    """loss.py:68-95 -- ProtoPNet separation cost."""

    def __init__(self, loss_weight, num_classes=4, reduction="mean"):
        self.loss_weight, self.num_classes, self.reduction = loss_weight, num_classes, reduction

    def compute(self, min_distances, target):
        if self.loss_weight == 0:
            return _zero(target)
        other = 1 - torch.nn.functional.one_hot(target, num_classes=self.num_classes)
        return -self.loss_weight * _batch_reduce(_per_class_extreme(min_distances, self.num_classes, largest=False) * other, self.reduction)


class ClusterRoiFeat(object):
    """loss.py:98-138 -- XProtoNet cluster cost on similarities (N, P)."""

    def __init__(self, loss_weight, num_classes=4, reduction="sum"):
        self.loss_weight, self.num_classes, self.reduction = loss_weight, num_classes, reduction

    def compute(self, similarities, target):
        if self.loss_weight == 0:
            return _zero(target)
        own = torch.nn.functional.one_hot(target, num_classes=self.num_classes)
        return -self.loss_weight * _batch_reduce(_per_class_extreme(similarities, self.num_classes, largest=True) * own, self.reduction)


class SeparationRoiFeat(object):
    """loss.py:141-183 -- XProtoNet separation cost; the abstain class (last) is never penalised when ``abstain_class``."""

    def __init__(self, loss_weight, num_classes=4, reduction="sum", abstain_class=True):
        self.loss_weight, self.num_classes, self.reduction, self.abstain_class = loss_weight, num_classes, reduction, abstain_class

    def compute(self, similarities, target):
        if self.loss_weight == 0:
            return _zero(target)
        own = torch.nn.functional.one_hot(target, num_classes=self.num_classes)
        if self.abstain_class:
            own = own.clone()
            own[:, -1] = 1
        return self.loss_weight * _batch_reduce(_per_class_extreme(similarities, self.num_classes, largest=True) * (1 - own), self.reduction)


class OrthogonalityLoss(object):
    """loss.py:186-229 -- sum of the pairwise cosine similarities of the prototypes (upper triangle), per class or over all."""

    def __init__(self, loss_weight, num_classes=4, mode="per_class"):
        if mode not in ("per_class", "all"):
            raise ValueError("mode must be 'per_class' or 'all'")
        self.loss_weight, self.num_classes, self.mode = loss_weight, num_classes, mode

    def compute(self, prototype_vectors):
        if self.loss_weight == 0:
            return _zero(prototype_vectors)
        p = prototype_vectors.reshape(prototype_vectors.shape[0], prototype_vectors.shape[1])
        if self.mode == "per_class":
            p = p.reshape(self.num_classes, -1, p.shape[1])
            sim = torch.nn.functional.cosine_similarity(p.unsqueeze(1), p.unsqueeze(2), dim=3)
        else:
            sim = torch.nn.functional.cosine_similarity(p.unsqueeze(1), p.unsqueeze(0), dim=2)
        return self.loss_weight * torch.triu(sim, diagonal=1).sum()


class L_norm(object):  # noqa: N801 -- reference class name
    """loss.py:232-254 -- p-norm of a tensor (the occurrence maps), optionally masked."""

    def __init__(self, mask=None, p=1, loss_weight=1e-4, reduction="sum"):
        self.mask, self.p, self.loss_weight, self.reduction = mask, p, loss_weight, reduction

    def compute(self, tensor, dim=None):
        if self.loss_weight == 0:
            return _zero(tensor)
        t = tensor if self.mask is None else self.mask.to(tensor.device) * tensor
        loss = t.norm(p=self.p, dim=dim)
        if self.reduction == "mean":
            loss = loss.mean(dim=0).sum()
        elif self.reduction == "sum":
            loss = loss.sum()
        return self.loss_weight * loss


class CeLossAbstain(object):
    """loss.py:323-371 -- cross entropy with a learned abstention output (the K+1-th logit)."""

    def __init__(self, loss_weight=1, ab_weight=0.3, reduction="sum", ab_logitpath="joined"):
        if ab_logitpath not in ("joined", "separate"):
            raise AssertionError("ab_logitpath must be 'joined' or 'separate'")
        self.loss_weight, self.ab_weight, self.reduction, self.ab_logitpath = loss_weight, ab_weight, reduction, ab_logitpath

    def to(self, device):
        return None

    def compute(self, logits, target):
        if self.loss_weight == 0:
            return _zero(target)
        k = logits.shape[1] - 1
        assert k >= 2, "CeLossAbstain input must have >= 2 classes not including abstention"
        abstain = (logits.softmax(dim=1) if self.ab_logitpath == "joined" else logits.sigmoid())[:, k: k + 1]
        virtual = (1 - abstain) * logits[:, :k].softmax(dim=1) + abstain * torch.nn.functional.one_hot(target, num_classes=k)
        loss_pred = torch.nn.functional.nll_loss(torch.log(virtual), target, reduction=self.reduction)
        loss_abs = -torch.log(1 - abstain).squeeze()
        loss_abs = loss_abs.mean() if self.reduction == "mean" else loss_abs.sum() if self.reduction == "sum" else loss_abs
        return self.loss_weight * (loss_pred + self.ab_weight * loss_abs)
