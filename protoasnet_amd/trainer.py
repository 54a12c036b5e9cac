"""Data-parallel training harness: the reference's agent loop around the compiled training step.

What it reproduces, with the reference lines it follows:

* optimizer groups and learning rates -- ``lr_same`` (all parameters, ``weight_decay=1e-3``) or ``lr_disjoint`` (trunk / add-on /
  occurrence module with weight decay, prototypes and last layer without): ``src/agents/XProtoNet_e2e.py:36-82``;
* the loss recipe and its classes: ``XProtoNet_Base.py:54-81`` + ``Video_XProtoNet_e2e.py:86-110`` (``protoasnet_amd.losses`` has the
  classes with the reference's names and ``compute`` signatures);
* gradient accumulation WITHOUT dividing the loss, ``optimizer.step()`` every ``accumulation_steps`` micro-batches:
  ``Video_XProtoNet_e2e.py:137-142``;
* ``ReduceLROnPlateau`` stepped on the validation mean F1 (``StepLR`` stepped unconditionally), the warm push at
  ``num_warm_epochs`` without replacement, the push every ``push_rate`` epochs from ``push_start`` followed by a ``val_push`` epoch:
  ``XProtoNet_e2e.py:110-148``;
* the checkpoint dictionary ``{epoch, iteration, state_dict, optimizer}`` and the files ``last.pth`` / ``model_best.pth`` /
  ``epoch_<k>.pth``: ``src/agents/base.py:143-169``, ``XProtoNet_e2e.py:84-107``.

What it changes, because it has to scale:

* **one process per GPU**; every rank iterates its own loader shard; parameter gradients are SUMMED over ranks in ONE all-reduce of
  the flat bucket at each accumulation boundary (``dp.allreduce_gradients``), and each rank accumulates ``accumulation_steps / world``
  micro-batches -- so W ranks x k micro-batches take exactly the optimizer step of one process with W*k micro-batches (tested);
* **no per-batch host synchronisation**: the reference calls ``.item()`` seven times and runs sklearn on every batch
  (``Video_XProtoNet_e2e.py:112-135,143-153``); here predictions feed a confusion matrix that stays on the device, the loss terms
  accumulate in a device vector, and both come to the host once per epoch (after one tiny all-reduce across ranks, so every rank
  takes the same scheduler / best-model decisions);
* ``TransformLoss`` is skipped when its ``loss_weight`` is 0 (the reference returns before the second trunk pass too: loss.py:284);
  in ``val`` / ``val_push`` epochs -- where the reference computes the term as well (``Video_XProtoNet_e2e.py:72,98``) -- the warped clips
  ride in the SAME forward as the originals (one 2N-clip launch list; exact in eval mode, where the norm layers use running statistics);
* **every rank holds the same model**: parameters and buffers are broadcast from rank 0 at construction; the norm layers' running
  estimates (updated per rank from its own micro-batches, as under plain BatchNorm) are averaged over the ranks before every
  evaluation epoch, push and checkpoint, so all ranks evaluate, project and save ONE model;
* every rank must bring the same number of micro-batches per epoch (checked with one MIN/MAX all-reduce at the start of the epoch: a
  rank with fewer would issue fewer gradient all-reduces and hang the job); gradients of a trailing partial accumulation are kept
  across the epoch boundary, as the reference keeps them (it never zeroes at the start of an epoch).

Out of scope: wandb / CSV logging, AUC, the diversity counters with their hard-coded ``[:30]`` split, plots.
"""
from __future__ import annotations

import logging
import os
from copy import deepcopy
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _lib, dp, losses
from . import push as push_mod


def confusion_to_metrics(cm: torch.Tensor) -> Dict[str, object]:
    """Balanced accuracy and per-class F1 (``zero_division=0``) from a (K, K) confusion matrix [true, predicted] -- what
    ``sklearn.metrics.balanced_accuracy_score`` / ``f1_score(average=None, labels=range(K))`` return (Video_XProtoNet_e2e.py:244-252)."""
    cm = cm.double()
    tp = cm.diag()
    support, predicted = cm.sum(1), cm.sum(0)
    recall = torch.where(support > 0, tp / support.clamp(min=1), torch.zeros_like(tp))
    present = support > 0
    accu = float(recall[present].mean()) if bool(present.any()) else 0.0  # classes absent from y_true do not count
    denom = support + predicted
    f1 = torch.where(denom > 0, 2 * tp / denom.clamp(min=1), torch.zeros_like(tp))
    return {"accuracy": accu, "f1": f1.tolist(), "f1_mean": float(f1.mean())}


class DPTrainer:
    """``config`` carries the reference's keys: ``abstain_class``, ``save_dir`` and ``train`` (``num_train_epochs``,
    ``num_warm_epochs``, ``accumulation_steps``, ``push_start``, ``push_rate``, ``save``, ``save_step``, ``criterion``, ``optimizer``,
    ``lr_schedule``) as in ``src/configs/Ours_ProtoASNet_Video.yml:19-74``.  ``data_loaders`` maps ``train`` / ``val`` /
    ``train_push`` to iterables of ``{"cine", "target_AS", "filename"}`` batches -- already sharded per rank for train / val (a
    ``DistributedSampler`` or disjoint file lists); the push loader is the FULL loader (the sweep shards it itself)."""

    def __init__(self, model: torch.nn.Module, config: dict, data_loaders: Dict[str, object], rank: int = 0, world_size: int = 1,
                 log=logging.info):
        self.model, self.config, self.train_config = model, config, config["train"]
        self.data_loaders, self.rank, self.world_size, self.log = data_loaders, rank, world_size, log
        acc = int(self.train_config.get("accumulation_steps", 1))
        if acc % world_size != 0:
            raise ValueError(f"accumulation_steps={acc} must be a multiple of the world size {world_size}: every rank accumulates "
                             "accumulation_steps / world micro-batches before the one gradient all-reduce")
        if world_size > 1 and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("DPTrainer(world_size > 1) needs an initialised process group (one process per GPU)")
        self.local_accumulation = acc // world_size
        self.device = next(model.parameters()).device
        self.current_epoch, self.current_iteration, self.best_metric = 0, 0, 0.0
        self.num_real_classes = model.num_classes - 1 if config.get("abstain_class") else model.num_classes
        self.get_criterion()
        self.get_optimizer()
        self.scheduler = self.get_lr_scheduler()
        self.params = [p for g in self.optimizer.param_groups for p in g["params"]]
        self.sync_model_state()  # ranks built from different seeds / checkpoints would otherwise stay different models for ever

    # ---- one model on every rank ------------------------------------------------------------------------------------------------
    def sync_model_state(self) -> None:
        """Broadcast every parameter and buffer from rank 0 (what DistributedDataParallel does at construction)."""
        if self.world_size <= 1:
            return
        with torch.no_grad():
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                buf = push_mod._for_collective(t.detach())
                dist.broadcast(buf, src=0)
                if buf.data_ptr() != t.data_ptr():  # staged through the host (gloo rehearsal): bring it back; RCCL wrote t itself
                    t.copy_(buf.to(t.device))
        if hasattr(self.model, "cnn_backbone") and hasattr(self.model.cnn_backbone, "invalidate_plans"):
            self.model.cnn_backbone.invalidate_plans()  # a broadcast straight into the storage bumps no version counter
        self._norm_dirty = False

    def sync_norm_buffers(self) -> None:
        """Average the norm layers' running means / variances over the ranks and take the largest batch counter: each rank's estimates
        saw only its own micro-batches; evaluation, push and the checkpoint must use ONE model.

        Only when a training epoch ran since the last sync (``_norm_dirty``; every rank runs the same epochs, so the flag agrees
        across ranks and the collective stays matched): a sync of already-equal buffers would still ``copy_`` into every buffer, bump
        its version counter and make the next eval forward re-plan and re-pack every weight -- and for world sizes that are not powers
        of two ``(a + a + a) / 3`` need not be ``a``, so the statistics would drift by an ulp per call.  Only norm statistics are
        exchanged: ``prototype_class_identity``-like constants and ``ones`` are equal by construction."""
        if self.world_size <= 1:
            return
        norms = [m for m in self.model.modules() if isinstance(m, torch.nn.modules.batchnorm._NormBase) and m.running_mean is not None]
        # The decision to skip is COLLECTIVE (round 5, ADVICE): the local flag only knows about run_epoch("train"); a train-mode forward
        # outside it, or a per-rank load_state_dict, also moves the statistics.  One 3-word MAX-reduce: [any rank dirty, max, -min] of the
        # ranks' summed batch counters -- a sync runs when any rank trained since the last one or the counters disagree; every rank then
        # takes the same branch, so the collectives below stay matched.
        nbt = sum(int(m.num_batches_tracked) for m in norms if m.num_batches_tracked is not None)
        probe = torch.tensor([int(bool(getattr(self, "_norm_dirty", True))), nbt, -nbt], dtype=torch.int64, device=self.device)
        probe = push_mod._for_collective(probe)
        dist.all_reduce(probe, op=dist.ReduceOp.MAX)
        if int(probe[0]) == 0 and int(probe[1]) == -int(probe[2]):
            return
        fl = [b for m in norms for b in (m.running_mean, m.running_var)]
        it = [m.num_batches_tracked for m in norms if m.num_batches_tracked is not None]
        with torch.no_grad():
            if fl:
                flat = push_mod._for_collective(torch.cat([b.detach().float().flatten() for b in fl]))
                dist.all_reduce(flat)
                flat = (flat / self.world_size).to(self.device)
                o = 0
                for b in fl:
                    b.copy_(flat[o:o + b.numel()].view_as(b).to(b.dtype))
                    o += b.numel()
            if it:
                flat = push_mod._for_collective(torch.cat([b.detach().flatten().to(torch.int64) for b in it]))
                dist.all_reduce(flat, op=dist.ReduceOp.MAX)
                flat = flat.to(self.device)
                o = 0
                for b in it:
                    b.copy_(flat[o:o + b.numel()].view_as(b).to(b.dtype))
                    o += b.numel()
        self._norm_dirty = False

    # ---- reference surface: XProtoNet_Base.py:54-81 ---------------------------------------------------------------------------
    def get_criterion(self) -> None:
        cfg = deepcopy(self.train_config["criterion"])
        K = self.model.num_classes
        if self.config.get("abstain_class"):
            self.CeLoss = losses.CeLossAbstain(**cfg["CeLossAbstain"])
        else:
            self.CeLoss = losses.CeLoss(**cfg["CeLoss"])
        self.Cluster = losses.ClusterRoiFeat(num_classes=K, **cfg["ClusterRoiFeat"])
        self.Separation = losses.SeparationRoiFeat(num_classes=K, **cfg["SeparationRoiFeat"], abstain_class=bool(self.config.get("abstain_class")))
        self.Orthogonality = losses.OrthogonalityLoss(num_classes=K, **cfg["OrthogonalityLoss"])
        self.Lnorm_occurrence = losses.L_norm(**cfg["Lnorm_occurrence"])
        self.Trans_occurrence = losses.TransformLoss(**cfg["trans_occurrence"])
        self.Lnorm_fc = losses.L_norm(**cfg["Lnorm_FC"], mask=1 - torch.t(self.model.prototype_class_identity))

    # ---- XProtoNet_e2e.py:36-82 --------------------------------------------------------------------------------------------------
    def get_optimizer(self) -> None:
        cfg = deepcopy(self.train_config["optimizer"])
        name, mode = cfg.pop("name"), cfg.pop("mode")
        m = self.model
        if mode == "lr_same":
            specs = [{"params": m.parameters(), "lr": cfg["lr_same"], "weight_decay": 1e-3}]
        elif mode == "lr_disjoint":
            lr = cfg["lr_disjoint"]
            specs = [
                {"params": m.cnn_backbone.parameters(), "lr": lr["cnn_backbone"], "weight_decay": 1e-3},
                {"params": m.add_on_layers.parameters(), "lr": lr["add_on_layers"], "weight_decay": 1e-3},
                {"params": m.occurrence_module.parameters(), "lr": lr["occurrence_module"], "weight_decay": 1e-3},
                {"params": m.prototype_vectors, "lr": lr["prototype_vectors"]},
                {"params": m.last_layer.parameters(), "lr": lr["last_layer"]},
            ]
        else:
            raise ValueError(f"optimizer mode {mode} not valid.")
        self.optimizer = torch.optim.__dict__[name](specs)

    def get_lr_scheduler(self):
        cfg = deepcopy(self.train_config["lr_schedule"])
        name = cfg.pop("name")
        cfg.pop("verbose", None)  # accepted by the torch 1.13 the reference pins, gone from current torch
        return torch.optim.lr_scheduler.__dict__[name](self.optimizer, **cfg)

    # ---- base.py:143-169, XProtoNet_e2e.py:84-107 ------------------------------------------------------------------------------
    def get_state(self) -> dict:
        return {"epoch": self.current_epoch, "iteration": self.current_iteration, "state_dict": self.model.state_dict(),
                "optimizer": self.optimizer.state_dict()}

    def save_checkpoint(self, is_best: bool = False) -> None:
        if not self.train_config.get("save", True):
            return
        self.sync_norm_buffers()  # a collective: every rank takes part before rank 0 writes
        if self.rank != 0:
            return
        state, d = self.get_state(), self.config["save_dir"]
        os.makedirs(d, exist_ok=True)
        step = self.train_config.get("save_step")
        if step is not None and self.current_epoch % step == 0:
            torch.save(state, os.path.join(d, f"epoch_{self.current_epoch}.pth"))
        if is_best:
            torch.save(state, os.path.join(d, "model_best.pth"))
        torch.save(state, os.path.join(d, "last.pth"))

    def load_checkpoint(self, file_name: Optional[str]) -> bool:
        if file_name is None or not os.path.exists(file_name):
            self.log(f"No checkpoint exists from '{file_name}'. Skipping...")
            return False
        ck = torch.load(file_name, map_location=self.device)
        self.current_epoch, self.current_iteration = ck["epoch"], ck["iteration"]
        self.model.load_state_dict(ck["state_dict"])
        self.optimizer.load_state_dict(ck["optimizer"])
        self.log(f"Checkpoint loaded successfully from '{file_name}' at (epoch {ck['epoch']}) at (iteration {ck['iteration']})")
        return True

    # ---- Video_XProtoNet_e2e.py:36-361, minus the per-batch host work ------------------------------------------------------------
    def compute_loss(self, inp, target, logit, similarities, occurrence_map, occurrence_map_transformed=None, affine_config=None):
        terms = [
            self.CeLoss.compute(logits=logit, target=target),
            self.Cluster.compute(similarities, target),
            self.Separation.compute(similarities, target),
            self.Orthogonality.compute(self.model.prototype_vectors),
            self.Lnorm_occurrence.compute(occurrence_map, dim=(-3, -2, -1) if occurrence_map.dim() == 6 else (-2, -1)),
        ]
        if self.Trans_occurrence.loss_weight == 0:
            terms.append(torch.zeros((), device=logit.device))
        elif occurrence_map_transformed is not None:  # eval epochs: the warped clips rode in the same forward (run_epoch)
            terms.append(self.Trans_occurrence.compute_from_maps(occurrence_map, occurrence_map_transformed, affine_config))
        else:
            terms.append(self.Trans_occurrence.compute(inp, occurrence_map, self.model))
        terms.append(self.Lnorm_fc.compute(self.model.last_layer.weight))
        return sum(terms), torch.stack([t.detach().float().reshape(()) for t in terms])

    def run_epoch(self, epoch: int, mode: str = "train") -> Dict[str, object]:
        self.model.train() if mode == "train" else self.model.eval()
        loader = self.data_loaders[mode.split("_")[0] if "_push" in mode else mode]
        K = self.num_real_classes
        cm = torch.zeros(K * K, dtype=torch.int64, device=self.device)
        loss_sum = torch.zeros(7, dtype=torch.float32, device=self.device)
        n_batches = 0
        if self.world_size > 1:
            if hasattr(loader, "__len__"):  # a rank with fewer micro-batches would issue fewer collectives: the job would hang in RCCL
                n = torch.tensor([len(loader), -len(loader)], dtype=torch.int64, device=self.device)
                n = push_mod._for_collective(n)
                dist.all_reduce(n, op=dist.ReduceOp.MAX)
                if int(n[0]) != -int(n[1]):
                    raise RuntimeError(f"{mode} epoch {epoch}: the ranks' loaders hold between {-int(n[1])} and {int(n[0])} batches; every rank "
                                       "must bring the same number (shard with a DistributedSampler(drop_last=True) or equal file lists)")
            if mode != "train":
                self.sync_norm_buffers()
        # the reference's second trunk pass (loss.py:302) rides in the first one's launch list: eval epochs on running statistics, training
        # epochs with two statistics groups (model.forward_pair; PASN_NO_TRAIN_PAIR=1: two passes)
        warp_in_batch = self.Trans_occurrence.loss_weight != 0 and (mode != "train" or (hasattr(self.model, "forward_pair") and _lib.tuning_get("PASN_NO_TRAIN_PAIR") != "1"))
        with torch.set_grad_enabled(mode == "train"):
            for i, sample in enumerate(loader):
                inp = sample["cine"].to(self.device, non_blocking=True)
                target = sample["target_AS"].to(self.device, non_blocking=True)
                if warp_in_batch:  # the reference's second trunk pass (loss.py:302) shares the launch list of the first
                    cfg = losses.get_affine_config()
                    nb = inp.shape[0]
                    warped = losses.affine_warp(inp, cfg["angle"], cfg["scale"])
                    if hasattr(self.model, "forward_pair"):
                        (logit, similarities, occurrence_map), occ_t = self.model.forward_pair(inp, warped)
                    else:
                        logit, similarities, occurrence_map = self.model(torch.cat([inp, warped]))
                        occ_t = occurrence_map[nb:]
                        logit, similarities, occurrence_map = logit[:nb], similarities[:nb], occurrence_map[:nb]
                    loss, terms = self.compute_loss(inp, target, logit, similarities, occurrence_map, occ_t, cfg)
                else:
                    logit, similarities, occurrence_map = self.model(inp)
                    loss, terms = self.compute_loss(inp, target, logit, similarities, occurrence_map)
                pred = logit[:, :K].argmax(dim=1)  # softmax is monotone: the class of the largest real-class logit
                cm += torch.bincount(target.clamp(0, K - 1) * K + pred, minlength=K * K)
                loss_sum += terms
                n_batches += 1
                if mode == "train":
                    self._norm_dirty = True  # this rank's running statistics moved on their own
                    loss.backward()  # undivided, as the reference accumulates it
                    if (i + 1) % self.local_accumulation == 0:
                        dp.allreduce_gradients(self.params, average=False)  # ONE exchange per optimizer step: SUM over ranks
                        self.optimizer.step()
                        self.optimizer.zero_grad(set_to_none=True)
                    self.current_iteration += 1
        stats = torch.cat([cm.float(), loss_sum, torch.tensor([float(n_batches)], device=self.device)])
        if self.world_size > 1:  # one tiny all-reduce per epoch: every rank sees the global confusion matrix and takes the same decisions
            stats = push_mod._for_collective(stats)
            dist.all_reduce(stats)
        stats = stats.cpu()
        metrics = confusion_to_metrics(stats[: K * K].view(K, K))
        nb = max(float(stats[-1]), 1.0)
        metrics["loss_terms"] = (stats[K * K: K * K + 7] / nb).tolist()
        metrics["loss"] = float(sum(metrics["loss_terms"]))
        self.log(f"Epoch: {epoch} | {mode} | loss {metrics['loss']:.4f} | acc {metrics['accuracy']:.2%} | f1 {metrics['f1_mean']:.3f}")
        return metrics

    def push(self, replace_prototypes: bool = True):
        abstain = bool(self.config.get("abstain_class"))
        self.sync_norm_buffers()  # every shard of the sweep must see the same eval-mode statistics
        return push_mod.push_prototypes(
            self.data_loaders["train_push"], self.model, class_specific=True, abstain_class=abstain,
            root_dir_for_saving_prototypes=self.config.get("save_dir") and os.path.join(self.config["save_dir"], "img"),
            epoch_number=self.current_epoch, log=self.log, replace_prototypes=replace_prototypes, rank=self.rank, world_size=self.world_size)

    def save_model_w_condition(self, model_name: str, metric_dict: dict, threshold: float) -> None:
        name, metric = next(iter(metric_dict.items()))
        if metric > threshold and self.rank == 0 and self.train_config.get("save", True):
            os.makedirs(self.config["save_dir"], exist_ok=True)
            torch.save(self.get_state(), os.path.join(self.config["save_dir"], f"{model_name}_{name}-{metric:.4f}.pth"))

    def train(self) -> Dict[str, list]:
        tc = self.train_config
        history = {"train": [], "val": [], "val_push": []}
        for epoch in range(self.current_epoch, tc["num_train_epochs"]):
            self.current_epoch = epoch
            history["train"].append(self.run_epoch(epoch, mode="train"))
            val = self.run_epoch(epoch, mode="val")
            history["val"].append(val)
            if tc["lr_schedule"]["name"] == "StepLR":
                self.scheduler.step()
            else:
                self.scheduler.step(val["f1_mean"])
            if epoch == tc["num_warm_epochs"]:
                self.push(replace_prototypes=False)
            if epoch >= tc["push_start"] and epoch % tc["push_rate"] == 0:
                self.push()
                vp = self.run_epoch(epoch, mode="val_push")
                history["val_push"].append(vp)
                self.save_model_w_condition(model_name=f"{epoch}push", metric_dict={"f1": vp["f1_mean"]}, threshold=0.65)
                is_best = vp["f1_mean"] > self.best_metric
                if is_best:
                    self.best_metric = vp["f1_mean"]
                    self.log(f"achieved best model with mean_f1 of {vp['f1_mean']}")
                self.save_checkpoint(is_best=is_best)
            self.save_checkpoint(is_best=False)
        return history
