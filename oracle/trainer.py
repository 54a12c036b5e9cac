"""Oracle of the reference's TRAINING STEP and epoch loop (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates, over a ``state_dict`` of leaf tensors, what the reference's agents do around the hot path:

* ``Video_XProtoNet_e2e.run_epoch`` (src/agents/Video_XProtoNet_e2e.py:36-172): per micro-batch ``model(input)``, the seven loss
  terms in the order CE, cluster, separation, orthogonality, occurrence-map norm, occurrence-map transform, last-layer norm
  (:86-110), their UNDIVIDED sum, ``loss.backward()`` and ``optimizer.step(); optimizer.zero_grad()`` every
  ``accumulation_steps`` micro-batches (:137-142); in ``val`` / ``val_push`` epochs the same forward and the same seven terms in eval
  mode under ``no_grad`` (:72, :98 -- the transform term IS computed there too, on ``model.eval()``);
* ``XProtoNet_e2e.get_optimizer`` (src/agents/XProtoNet_e2e.py:22-64): ``lr_same`` = every parameter in one group with
  ``weight_decay=1e-3``; ``lr_disjoint`` = trunk / add-on / occurrence module with weight decay, prototype vectors and last layer
  without;
* the criterion set of ``XProtoNet_Base.get_criterion`` (src/agents/XProtoNet_Base.py:54-81), incl. the last-layer mask
  ``1 - prototype_class_identity^T`` for ``Lnorm_FC``.

The forward is the oracle's train-/eval-mode restatement (oracle/nets.py), the loss terms are oracle/losses.py (pinned by g6), the
optimizer is ``torch.optim`` itself (third-party in the reference as well).  Pin status: the ResNet-18 train-mode forward + backward
is pinned by the reference's own run (g7); the X3D / R(2+1)D trunks are parity-unpinned as everywhere; the affine
warp inside the transform term restates torchvision 0.14 (unpinned: torchvision is absent).

The random affine configuration is drawn exactly like ``get_affine_config`` (src/loss/loss.py:257-269): ``random.uniform(-20, 20)``
then ``random.uniform(0.6, 1.5)`` from Python's global generator, once per ``TransformLoss.compute`` call with a non-zero weight."""
from __future__ import annotations

import random
from copy import deepcopy
from typing import Dict, Iterable, List, Mapping

import torch

from . import heads, losses, nets

TERM_NAMES = ("ce", "cluster", "separation", "orthogonality", "occurrence_norm", "occurrence_transform", "fc_norm")


def get_affine_config() -> dict:
    """src/loss/loss.py:257-269 (the two keys that vary; translate 0, shear 0, fill 0, bilinear are fixed)."""
    angle = random.uniform(-20, 20)
    scale = random.uniform(0.6, 1.5)
    return {"angle": angle, "scale": scale}


def parameter_groups(sd: Mapping[str, torch.Tensor], opt_cfg: dict) -> List[dict]:
    """src/agents/XProtoNet_e2e.py:28-62 over state-dict names (``model.parameters()`` = every key that is not a norm buffer)."""
    params = [(k, v) for k, v in sd.items() if v.is_floating_point() and "running_" not in k and "num_batches_tracked" not in k]
    if opt_cfg["mode"] == "lr_same":
        return [{"params": [v for _, v in params], "lr": opt_cfg["lr_same"], "weight_decay": 1e-3}]
    if opt_cfg["mode"] != "lr_disjoint":
        raise ValueError(f"optimizer mode {opt_cfg['mode']} not valid.")
    lr = opt_cfg["lr_disjoint"]
    pick = lambda prefix: [v for k, v in params if k.startswith(prefix)]
    return [
        {"params": pick("cnn_backbone."), "lr": lr["cnn_backbone"], "weight_decay": 1e-3},
        {"params": pick("add_on_layers."), "lr": lr["add_on_layers"], "weight_decay": 1e-3},
        {"params": pick("occurrence_module."), "lr": lr["occurrence_module"], "weight_decay": 1e-3},
        {"params": [sd["prototype_vectors"]], "lr": lr["prototype_vectors"]},
        {"params": pick("last_layer."), "lr": lr["last_layer"]},
    ]


class ReferenceTrainer:
    """The reference's agent reduced to what touches the hot path.  ``sd``: reference-named state dict (CPU fp32); parameters are
    turned into leaves (``ones`` keeps ``requires_grad=False`` as in ProtoPNet.py:136)."""

    def __init__(self, sd: Mapping[str, torch.Tensor], train_config: dict, arch: str, num_classes: int, abstain_class: bool = False,
                 last_layer_num: int = -3):
        self.arch, self.num_classes, self.abstain, self.last_layer_num = arch, num_classes, abstain_class, last_layer_num
        self.cfg = deepcopy(train_config)
        self.sd: Dict[str, torch.Tensor] = {}
        for k, v in sd.items():
            t = v.detach().clone()
            if t.is_floating_point() and "running_" not in k and k != "ones":
                t.requires_grad_(True)
            self.sd[k] = t
        opt = deepcopy(self.cfg["optimizer"])
        self.optimizer = torch.optim.__dict__[opt["name"]](parameter_groups(self.sd, opt))
        P = self.sd["prototype_vectors"].shape[0]
        self.class_identity = heads.prototype_class_identity(P, num_classes)  # ProtoPNet.py:326-340
        self.fc_mask = 1 - torch.t(self.class_identity)                       # XProtoNet_Base.py:81
        self.iteration = 0

    # ---- Video_XProtoNet_e2e.py:84-110 -----------------------------------------------------------------------------------------
    def loss_terms(self, x, target, out, train: bool):
        c, K, sd = self.cfg["criterion"], self.num_classes, self.sd
        logit, sim, occ = out["logits"], out["similarity"], out["occurrence_map"]
        if self.abstain:
            ce = losses.ce_loss_abstain(logit, target, **c["CeLossAbstain"])
        else:
            ce = losses.ce_loss(logit, target, **c["CeLoss"])
        terms = [
            ce,
            losses.cluster_roi_feat(sim, target, num_classes=K, **c["ClusterRoiFeat"]),
            losses.separation_roi_feat(sim, target, num_classes=K, abstain_class=self.abstain, **c["SeparationRoiFeat"]),
            losses.orthogonality(sd["prototype_vectors"], num_classes=K, **c["OrthogonalityLoss"]),
            losses.l_norm(occ, dim=(-3, -2, -1) if x.dim() == 5 else (-2, -1), **c["Lnorm_occurrence"]),
        ]
        tc = c["trans_occurrence"]
        if tc["loss_weight"] == 0:
            terms.append(torch.tensor(0))  # loss.py:284-285
        else:
            cfg = get_affine_config()
            if train:
                occ_of = lambda xt: nets.xprotonet_train_forward(sd, xt, self.arch, self.last_layer_num, occurrence_only=True)["occurrence_map"]
            else:
                occ_of = lambda xt: nets.compute_occurence_map(sd, xt, self.arch, self.last_layer_num)
            terms.append(losses.transform_loss(x, occ, occ_of, cfg["angle"], cfg["scale"], tc["loss_weight"], tc.get("reduction", "sum")))
        terms.append(losses.l_norm(sd["last_layer.weight"], mask=self.fc_mask, **c["Lnorm_FC"]))
        return terms

    def run_epoch(self, batches: Iterable[dict], mode: str = "train") -> dict:
        """One epoch over ``batches`` ({"cine", "target_AS"}); returns summed loss terms, predictions and the confusion matrix."""
        train = mode == "train"
        acc = int(self.cfg.get("accumulation_steps", 1))
        K = self.num_classes - 1 if self.abstain else self.num_classes
        total = torch.zeros(7, dtype=torch.float64)
        cm = torch.zeros(K, K, dtype=torch.int64)
        preds = []
        # (no zero_grad here: the reference zeroes only after a step, so gradients of a trailing partial accumulation carry into the
        # next epoch, Video_XProtoNet_e2e.py:139-141)
        with torch.set_grad_enabled(train):
            for i, sample in enumerate(batches):
                x, target = sample["cine"].float(), sample["target_AS"]
                if train:
                    out = nets.xprotonet_train_forward(self.sd, x, self.arch, self.last_layer_num)
                else:
                    out = nets.xprotonet_forward(self.sd, x, self.arch, self.last_layer_num, contract=True)
                terms = self.loss_terms(x, target, out, train)
                loss = sum(terms)  # undivided (:99-107)
                pred = out["logits"][:, :K].softmax(dim=1).max(dim=1)[1]
                preds.append(pred)
                for t_, p_ in zip(target.tolist(), pred.tolist()):
                    cm[t_, p_] += 1
                if train:
                    loss.backward()
                    if (i + 1) % acc == 0:
                        self.optimizer.step()
                        self.optimizer.zero_grad()
                    self.iteration += 1
                total += torch.tensor([float(t.detach()) for t in terms], dtype=torch.float64)
        return {"loss_terms_sum": total, "confusion": cm, "pred": torch.cat(preds) if preds else torch.zeros(0, dtype=torch.long)}
