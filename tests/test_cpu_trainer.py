"""CPU: the data-parallel training harness (protoasnet_amd/trainer.py) on a small pure-torch stand-in with the model surface the
harness touches (cnn_backbone / add_on_layers / occurrence_module / prototype_vectors / last_layer, forward -> (logits, similarity,
occurrence_map)).  The HIP models cannot run here (no GPU); the harness is model-agnostic, and the GPU suite drives it with the
real Video ProtoASNet (tests/test_gpu_trainer.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

TRAIN_CFG = {
    "num_train_epochs": 3, "save": True, "save_step": 2, "num_warm_epochs": 99, "accumulation_steps": 4, "push_start": 99, "push_rate": 5,
    "criterion": {
        "CeLoss": {"loss_weight": 1, "reduction": "mean"},
        "CeLossAbstain": {"loss_weight": 1, "ab_weight": 0.3, "ab_logitpath": "joined", "reduction": "mean"},
        "ClusterRoiFeat": {"loss_weight": 0.8, "reduction": "mean"},
        "SeparationRoiFeat": {"loss_weight": 0.08, "reduction": "mean"},
        "OrthogonalityLoss": {"loss_weight": 0.01, "mode": "per_class"},
        "Lnorm_occurrence": {"p": 2, "loss_weight": 1e-4, "reduction": "mean"},
        "trans_occurrence": {"loss_weight": 0.0, "reduction": "mean"},  # the warp runs on the GPU only
        "Lnorm_FC": {"p": 1, "loss_weight": 1e-4},
    },
    "optimizer": {"name": "Adam", "mode": "lr_same", "lr_same": 1e-3},
    "lr_schedule": {"name": "ReduceLROnPlateau", "mode": "max", "factor": 0.5, "patience": 0, "threshold": 1e-4, "cooldown": 0,
                    "min_lr": 1e-6, "verbose": True},
}


class Toy(nn.Module):
    def __init__(self, P=8, K=4, D=6):
        super().__init__()
        torch.manual_seed(0)
        self.cnn_backbone = nn.Sequential(nn.Conv3d(3, 8, 3, padding=1), nn.ReLU())
        self.add_on_layers = nn.Sequential(nn.Conv3d(8, D, 1))
        self.occurrence_module = nn.Sequential(nn.Conv3d(8, P, 1, bias=False))
        self.prototype_vectors = nn.Parameter(torch.rand(P, D, 1, 1, 1))
        self.last_layer = nn.Linear(P, K, bias=False)
        self.num_classes, self.num_prototypes, self.prototype_shape = K, P, (P, D, 1, 1, 1)
        self.prototype_class_identity = torch.zeros(P, K)
        for j in range(P):
            self.prototype_class_identity[j, j // (P // K)] = 1

    def forward(self, x):
        f = self.cnn_backbone(x)
        z = self.add_on_layers(f)
        occ = self.occurrence_module(f).abs()
        feats = torch.einsum("npthw,ndthw->npd", occ, z)
        sim = (F.cosine_similarity(feats, self.prototype_vectors.flatten(1)[None], dim=2) + 1) / 2
        return self.last_layer(sim), sim, occ.unsqueeze(2)


def _batches(seed, n, B=2):
    g = torch.Generator().manual_seed(seed)
    return [{"cine": torch.randn(B, 3, 2, 6, 6, generator=g), "target_AS": torch.randint(0, 3, (B,), generator=g), "filename": ["x"] * B}
            for _ in range(n)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from protoasnet_amd.trainer import DPTrainer

    all_b = _batches(7, 8)
    mine = all_b[rank::world]  # micro-batches 0,2,4,6 / 1,3,5,7: two optimizer steps of 4 global micro-batches each
    cfg = {"abstain_class": True, "save_dir": os.path.join(out_dir, f"rank{rank}"), "train": dict(TRAIN_CFG, num_train_epochs=1)}
    t = DPTrainer(Toy(), cfg, {"train": mine, "val": all_b[:2]}, rank=rank, world_size=world, log=lambda *_: None)
    assert t.local_accumulation == 2
    m = t.run_epoch(0, "train")
    torch.save({"state": t.model.state_dict(), "metrics": m, "iteration": t.current_iteration}, os.path.join(out_dir, f"dp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_times_k_equals_one_rank_times_2k(tmp_path):
    """2 ranks x 2 micro-batches per optimizer step == 1 rank x 4 micro-batches: same parameters after two optimizer steps (the
    gradient SUM over ranks at the accumulation boundary; losses undivided as in Video_XProtoNet_e2e.py:137-142), same epoch metrics
    on every rank (global confusion matrix)."""
    from protoasnet_amd.trainer import DPTrainer

    world, port = 2, _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    all_b = _batches(7, 8)
    # the single process sees the micro-batches in the order the two ranks consume them pairwise: (0,1,2,3) then (4,5,6,7)
    cfg = {"abstain_class": True, "save_dir": str(tmp_path / "single"), "train": dict(TRAIN_CFG, num_train_epochs=1)}
    t = DPTrainer(Toy(), cfg, {"train": all_b, "val": all_b[:2]}, log=lambda *_: None)
    assert t.local_accumulation == 4
    ms = t.run_epoch(0, "train")
    outs = [torch.load(os.path.join(tmp_path, f"dp{r}.pt")) for r in range(world)]
    for o in outs:
        for k, v in t.model.state_dict().items():
            assert torch.allclose(o["state"][k], v, atol=2e-6, rtol=1e-5), k
        assert o["metrics"]["f1"] == pytest.approx(ms["f1"]) and o["metrics"]["accuracy"] == pytest.approx(ms["accuracy"])
        assert o["metrics"]["loss"] == pytest.approx(ms["loss"], rel=1e-5)
        assert o["iteration"] == 4
    assert t.current_iteration == 8


class ToyNorm(Toy):
    """Toy with a batch-statistics norm in the trunk (running estimates are per-rank state) and a seed-dependent initialisation."""

    def __init__(self, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.cnn_backbone = nn.Sequential(nn.Conv3d(3, 8, 3, padding=1), nn.BatchNorm3d(8), nn.ReLU())
        with torch.no_grad():
            for p in self.parameters():
                p.add_(torch.randn_like(p) * 0.05)


def _sync_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from protoasnet_amd.trainer import DPTrainer

    all_b = _batches(11, 8)
    cfg = {"abstain_class": True, "save_dir": os.path.join(out_dir, "ck"), "train": dict(TRAIN_CFG, num_train_epochs=1)}
    t = DPTrainer(ToyNorm(seed=100 + rank), cfg, {"train": all_b[rank::world], "val": all_b[:2]}, rank=rank, world_size=world, log=lambda *_: None)
    after_init = {k: v.clone() for k, v in t.model.state_dict().items()}
    t.run_epoch(0, "train")
    before_sync = {k: v.clone() for k, v in t.model.state_dict().items() if "running" in k}
    t.run_epoch(0, "val")  # averages the norm buffers first
    versions = [b._version for b in t.model.buffers()]
    t.save_checkpoint()
    t.run_epoch(0, "val")  # nothing trained since the last sync: no collective, no copy_, no version bump (cached HIP plans stay valid)
    clean = versions == [b._version for b in t.model.buffers()] and not t._norm_dirty
    torch.save({"init": after_init, "before": before_sync, "final": t.model.state_dict(), "clean": clean}, os.path.join(out_dir, f"sync{rank}.pt"))
    # a rank with a shorter loader must be refused, not hang: both ranks take part in the length check
    t.data_loaders["train"] = all_b[: 2 if rank == 0 else 4]
    try:
        t.run_epoch(1, "train")
        raised = False
    except RuntimeError as e:
        raised = "same number" in str(e)
    torch.save(raised, os.path.join(out_dir, f"raised{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ranks_built_from_different_seeds_train_one_model(tmp_path):
    """ADVICE (round 2): ranks whose models were built from different seeds start from rank 0's parameters and buffers (broadcast at
    construction), take identical optimizer steps, evaluate / checkpoint with the SAME averaged running statistics, and a shard with
    fewer micro-batches is refused with an error on every rank instead of hanging the collectives."""
    world, port = 2, _free_port()
    mp.spawn(_sync_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(os.path.join(tmp_path, f"sync{r}.pt")) for r in range(world))
    ref0 = ToyNorm(seed=100).state_dict()
    for k, v in a["init"].items():
        assert torch.equal(v, b["init"][k]), f"{k}: ranks differ after construction"
        assert torch.equal(v, ref0[k]), f"{k}: not rank 0's value"
    assert any(not torch.equal(a["before"][k], b["before"][k]) for k in a["before"]), "the ranks saw different micro-batches"
    for k, v in a["final"].items():
        assert torch.allclose(v.float(), b["final"][k].float(), atol=1e-6), f"{k}: ranks differ after the epoch"
        if "running" in k and v.is_floating_point():
            assert torch.allclose(v, (a["before"][k] + b["before"][k]) / 2, atol=1e-6), f"{k}: not the mean of the ranks' estimates"
    assert os.path.exists(os.path.join(tmp_path, "ck", "last.pth"))
    assert a["clean"] and b["clean"], "a sync with nothing trained in between must not touch the buffers"
    assert all(torch.load(os.path.join(tmp_path, f"raised{r}.pt")) for r in range(world))


def test_sharded_push_refuses_a_shuffling_loader():
    """ADVICE (round 2): every rank re-iterates the batch sampler on its own; with a random sampler the shards would overlap."""
    from protoasnet_amd import push

    ds = torch.utils.data.TensorDataset(torch.zeros(8, 1))
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=True)
    with pytest.raises(ValueError, match="SequentialSampler"):
        next(push._iter_shard(loader, 0, 2))
    assert next(push._iter_shard(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False), 1, 2))[0] == 2


def test_optimizer_groups_scheduler_and_checkpoint_follow_the_reference(tmp_path):
    from protoasnet_amd.trainer import DPTrainer, confusion_to_metrics

    b = _batches(3, 4)
    cfg = {"abstain_class": True, "save_dir": str(tmp_path), "train": dict(TRAIN_CFG)}
    t = DPTrainer(Toy(), cfg, {"train": b, "val": b[:2]}, log=lambda *_: None)
    g = t.optimizer.param_groups
    assert len(g) == 1 and g[0]["lr"] == 1e-3 and g[0]["weight_decay"] == 1e-3 and type(t.optimizer).__name__ == "Adam"
    dis = dict(TRAIN_CFG, optimizer={"name": "Adam", "mode": "lr_disjoint", "lr_disjoint": {
        "cnn_backbone": 1e-4, "add_on_layers": 3e-3, "occurrence_module": 3e-3, "prototype_vectors": 3e-3, "last_layer": 1e-4}})
    t2 = DPTrainer(Toy(), {"abstain_class": False, "save_dir": str(tmp_path), "train": dis}, {"train": b, "val": b[:2]}, log=lambda *_: None)
    g2 = t2.optimizer.param_groups
    assert [x["lr"] for x in g2] == [1e-4, 3e-3, 3e-3, 3e-3, 1e-4]
    assert [x["weight_decay"] for x in g2] == [1e-3, 1e-3, 1e-3, 0, 0]  # prototypes and last layer carry no weight decay
    assert type(t2.CeLoss).__name__ == "CeLoss" and type(t.CeLoss).__name__ == "CeLossAbstain"
    hist = t.train()
    assert len(hist["train"]) == 3 and hist["train"][-1]["loss"] < hist["train"][0]["loss"]
    assert type(t.scheduler).__name__ == "ReduceLROnPlateau"
    ck = torch.load(tmp_path / "last.pth")
    assert sorted(ck) == ["epoch", "iteration", "optimizer", "state_dict"] and ck["epoch"] == 2 and ck["iteration"] == 12  # base.py:143-149
    assert (tmp_path / "epoch_0.pth").exists() and (tmp_path / "epoch_2.pth").exists() and not (tmp_path / "epoch_1.pth").exists()
    t3 = DPTrainer(Toy(), cfg, {"train": b, "val": b[:2]}, log=lambda *_: None)
    assert t3.load_checkpoint(str(tmp_path / "last.pth")) and t3.current_epoch == 2
    for k, v in t.model.state_dict().items():
        assert torch.equal(t3.model.state_dict()[k], v)
    assert not t3.load_checkpoint(str(tmp_path / "missing.pth"))
    # metrics == sklearn's on a hand case (balanced accuracy ignores absent classes; f1 zero_division=0)
    cm = torch.tensor([[2, 1, 0], [0, 3, 0], [0, 0, 0]])
    m = confusion_to_metrics(cm)
    assert m["accuracy"] == pytest.approx((2 / 3 + 1) / 2) and m["f1"] == pytest.approx([0.8, 6 / 7, 0.0])
    with pytest.raises(ValueError, match="multiple of the world size"):
        DPTrainer(Toy(), {"abstain_class": True, "save_dir": "", "train": dict(TRAIN_CFG, accumulation_steps=3)}, {}, world_size=2)


def test_oracle_reference_trainer_step_semantics():
    """``oracle.trainer.ReferenceTrainer`` (the checker of the GPU trainer-parity test) against the semantics of the reference step spelled
    out by hand: the losses of the ``accumulation_steps`` micro-batches are summed UNDIVIDED (Video_XProtoNet_e2e.py:99-107,137-141), the
    trunk / add-on / occurrence-module groups carry ``weight_decay=1e-3``, prototypes and last layer none (XProtoNet_e2e.py:38-62), the
    transform term draws one affine configuration per micro-batch (loss.py:257-269, 283-320), and norm statistics move twice per micro-batch
    (forward + ``compute_occurence_map`` of the warped clip).  X3D-S + head B on two tiny clips, SGD so that the update IS the gradient."""
    import random
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle
    from protoasnet_amd import synth
    from util import CFG_VIDEO_X3D, synth_model

    m = synth_model(CFG_VIDEO_X3D)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    lr = {"cnn_backbone": 1e-3, "add_on_layers": 3e-3, "occurrence_module": 3e-3, "prototype_vectors": 3e-3, "last_layer": 1e-5}
    tc = dict(TRAIN_CFG, accumulation_steps=2, optimizer={"name": "SGD", "mode": "lr_disjoint", "lr_disjoint": lr})
    tc["criterion"] = dict(tc["criterion"], trans_occurrence={"loss_weight": 1e-2, "reduction": "mean"})
    batches = [{"cine": synth.echo_clips((2, 3, 4, 64, 64), seed=10 + b), "target_AS": (torch.arange(2) + b) % 3} for b in range(2)]
    ref = oracle.trainer.ReferenceTrainer(sd0, tc, arch="x3d_s", num_classes=3)
    random.seed(5)
    out = ref.run_epoch(batches, "train")
    assert ref.iteration == 2 and float(out["loss_terms_sum"][5]) > 0

    # by hand: gradients of loss(batch 0) + loss(batch 1), statistics carried from one pass to the next, one SGD step
    sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running_" not in k and k != "ones" else v.clone()) for k, v in sd0.items()}
    random.seed(5)
    helper = oracle.trainer.ReferenceTrainer(sd0, tc, arch="x3d_s", num_classes=3)
    helper.sd = sd
    total = 0
    for b in batches:
        o = oracle.nets.xprotonet_train_forward(sd, b["cine"], "x3d_s")
        total = total + sum(helper.loss_terms(b["cine"], b["target_AS"], o, True))
    total.backward()
    for k, p0 in sd0.items():
        if not p0.is_floating_point():
            continue  # num_batches_tracked: the functional oracle does not keep the counter (momentum is fixed, nothing reads it)
        if "running_" in k:
            assert torch.allclose(ref.sd[k], sd[k], rtol=1e-6, atol=1e-7), k
            continue
        if k == "ones":
            assert torch.equal(ref.sd[k], p0)
            continue
        group = next(g for g in lr if k.startswith(g))
        wd = 1e-3 if group in ("cnn_backbone", "add_on_layers", "occurrence_module") else 0.0
        want = p0 - lr[group] * (sd[k].grad + wd * p0)
        assert torch.allclose(ref.sd[k].detach(), want, rtol=1e-5, atol=1e-8), k
