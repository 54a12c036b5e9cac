"""Seeded inputs and the case list shared by tests/golden/make_golden_losses.py (which runs the reference) and
tests/test_cpu_losses.py (which runs protoasnet_amd.losses on the same inputs)."""
import torch

N, K, P, D = 6, 4, 40, 16


def make_inputs(kind):
    g = torch.Generator().manual_seed({"scores": 1, "logits": 2, "protos": 3, "maps": 4}[kind])
    target = torch.tensor([0, 1, 2, 3, 1, 0])
    if kind == "scores":      # similarities / min_distances (N, P) + labels
        return torch.rand(N, P, generator=g), target
    if kind == "logits":      # (N, K [+1]) + labels over the K real classes
        return torch.randn(N, K + 1, generator=g), target
    if kind == "protos":      # prototype vectors (P, D, 1, 1, 1)
        return (torch.rand(P, D, 1, 1, 1, generator=g),)
    if kind == "maps":        # occurrence maps (N, P, 1, T, H, W)
        return (torch.rand(N, P, 1, 2, 3, 3, generator=g) - 0.3,)
    raise KeyError(kind)


CASES = [
    # tag, class, constructor kwargs, input kind
    ("ce_mean", "CeLoss", dict(loss_weight=0.7, reduction="mean"), "logits"),
    ("cluster_patch", "ClusterPatch", dict(loss_weight=0.8, num_classes=4, reduction="mean"), "scores"),
    ("cluster_patch_sum", "ClusterPatch", dict(loss_weight=0.8, num_classes=4, reduction="sum"), "scores"),
    ("sep_patch", "SeparationPatch", dict(loss_weight=0.08, num_classes=4, reduction="mean"), "scores"),
    ("cluster_roi", "ClusterRoiFeat", dict(loss_weight=0.8, num_classes=4, reduction="sum"), "scores"),
    ("cluster_roi_mean", "ClusterRoiFeat", dict(loss_weight=0.8, num_classes=4, reduction="mean"), "scores"),
    ("sep_roi_abstain", "SeparationRoiFeat", dict(loss_weight=0.08, num_classes=4, reduction="sum", abstain_class=True), "scores"),
    ("sep_roi_plain", "SeparationRoiFeat", dict(loss_weight=0.08, num_classes=4, reduction="mean", abstain_class=False), "scores"),
    ("ortho_per_class", "OrthogonalityLoss", dict(loss_weight=0.01, num_classes=4, mode="per_class"), "protos"),
    ("ortho_all", "OrthogonalityLoss", dict(loss_weight=0.01, num_classes=4, mode="all"), "protos"),
    ("l1_sum", "L_norm", dict(p=1, loss_weight=1e-2, reduction="sum"), "maps"),
    ("l2_mean", "L_norm", dict(p=2, loss_weight=1e-2, reduction="mean"), "maps"),
    ("ce_abstain_joined", "CeLossAbstain", dict(loss_weight=1.0, ab_weight=0.3, reduction="sum", ab_logitpath="joined"), "logits"),
    ("ce_abstain_separate", "CeLossAbstain", dict(loss_weight=0.5, ab_weight=0.1, reduction="mean", ab_logitpath="separate"), "logits"),
]
