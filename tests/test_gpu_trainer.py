"""GPU: the data-parallel training harness driving the real Video ProtoASNet (X3D-S trunk + head B, compiled forward + backward
launch lists): the reference's loss recipe incl. TransformLoss, accumulation, scheduler, warm push + push + val_push schedule,
checkpoint round trip through load_state_dict."""
import os

import pytest
import torch

from protoasnet_amd import synth
from test_cpu_trainer import TRAIN_CFG
from util import CFG_VIDEO_X3D, synth_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _loader(n, seed, B=2):
    class L(list):
        batch_size = B

    out = L()
    for b in range(n):
        out.append({"cine": synth.echo_clips((B, 3, 4, 64, 64), seed=seed + b), "target_AS": (torch.arange(B) + b) % 3,
                    "filename": [f"c{b}_{i}" for i in range(B)]})
    return out


@pytest.mark.timeout(900)
def test_trainer_runs_the_reference_schedule_on_the_hip_model(tmp_path):
    from protoasnet_amd.trainer import DPTrainer

    m = synth_model(CFG_VIDEO_X3D).to(DEV)
    tc = dict(TRAIN_CFG, num_train_epochs=3, num_warm_epochs=0, push_start=1, push_rate=1, accumulation_steps=2, save_step=1)
    tc["criterion"] = dict(tc["criterion"], trans_occurrence={"loss_weight": 1e-3, "reduction": "mean"})  # second differentiable trunk pass
    cfg = {"abstain_class": False, "save_dir": str(tmp_path), "train": tc}
    logs = []
    t = DPTrainer(m, cfg, {"train": _loader(4, 10), "val": _loader(2, 50), "train_push": _loader(4, 10)}, log=lambda s, *_: logs.append(str(s)))
    before = m.prototype_vectors.detach().clone()
    hist = t.train()
    assert len(hist["train"]) == 3 and len(hist["val"]) == 3 and len(hist["val_push"]) == 2  # pushes at epochs 1 and 2
    assert all(torch.isfinite(torch.tensor(h["loss"])) for h in hist["train"])
    assert hist["train"][-1]["loss_terms"][5] != 0.0  # TransformLoss ran (its second trunk pass)
    assert any("push at epoch 0" in s for s in logs) and any("push at epoch 1" in s for s in logs)  # warm push + scheduled push
    assert not torch.equal(m.prototype_vectors.detach(), before)  # prototypes were projected
    # pushed prototypes sit on features of the training set: distance of each prototype to its source clip ~ 0 was tested elsewhere;
    # here: checkpoint files and a bit-exact round trip through load_state_dict into a fresh model
    for f in ("last.pth", "epoch_0.pth", "epoch_2.pth"):
        assert os.path.exists(tmp_path / f)
    ck = torch.load(tmp_path / "last.pth")
    assert sorted(ck) == ["epoch", "iteration", "optimizer", "state_dict"] and ck["iteration"] == 12
    m2 = synth_model(CFG_VIDEO_X3D).to(DEV)
    t2 = DPTrainer(m2, cfg, {"train": _loader(4, 10), "val": _loader(2, 50)}, log=lambda *_: None)
    assert t2.load_checkpoint(str(tmp_path / "last.pth"))
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v), k
    x = _loader(1, 50)[0]["cine"].to(DEV)
    m.eval(), m2.eval()
    with torch.no_grad():
        assert torch.equal(m(x)[0], m2(x)[0])
    assert os.path.exists(tmp_path / "img" / "epoch-1" / "prototypes_info.pickle")  # push artefacts of the scheduled push


def _kink_sparse(m):
    """+2.5 on the bias of every norm that feeds a ReLU: 0.6 % of the units masked instead of half of them, so a handful of ReLU masks that
    flip between two fp32 implementations cannot move whole gradient tensors ("Training parity and ReLU kinks")."""
    relu_fed = {n + ".bias" for n, mod in m.named_modules() if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d))
                and not any(t in n for t in ("downsample", "shortcut", "bn_b"))}
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name in relu_fed:
                p += 2.5
    return m


@pytest.mark.timeout(900)
@pytest.mark.parametrize("opt", ["SGD", "Adam"])
def test_trainer_matches_the_oracle_restatement_of_the_reference_step(opt):
    """SURVEY 8f-2 parity: two optimizer steps of two micro-batches each through the HIP ``DPTrainer`` against ``oracle.trainer.
    ReferenceTrainer`` -- the reference's epoch loop (losses summed undivided, ``optimizer.step()`` every ``accumulation_steps``,
    parameter groups with / without weight decay: Video_XProtoNet_e2e.py:77-142, XProtoNet_e2e.py:38-62) restated over the oracle's
    train-mode forward and loss terms -- with the reference's loss recipe INCLUDING the transform term (the second differentiable trunk
    pass), then a validation epoch in eval mode (where the reference computes the transform term as well).  SGD makes the comparison
    sensitive to gradient MAGNITUDES (a first Adam step is lr * sign(g) whatever the size of g); Adam is the reference's optimizer."""
    import random

    import oracle
    from protoasnet_amd.trainer import DPTrainer

    m = _kink_sparse(synth_model(CFG_VIDEO_X3D)).to(DEV)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    lr = 2e-3 if opt == "SGD" else 1e-3
    tc = dict(TRAIN_CFG, num_train_epochs=1, accumulation_steps=2, num_warm_epochs=99, push_start=99,
              optimizer={"name": opt, "mode": "lr_disjoint",
                         "lr_disjoint": {"cnn_backbone": lr, "add_on_layers": 3 * lr, "occurrence_module": 3 * lr, "prototype_vectors": 3 * lr,
                                         "last_layer": 1e-2 * lr}})
    tc["criterion"] = dict(tc["criterion"], trans_occurrence={"loss_weight": 1e-2, "reduction": "mean"},
                           Lnorm_occurrence={"p": 2, "loss_weight": 1e-3, "reduction": "mean"})
    cfg = {"abstain_class": False, "save_dir": None, "train": dict(tc, save=False)}
    train, val = _loader(4, 10), _loader(2, 50)
    t = DPTrainer(m, cfg, {"train": train, "val": val}, log=lambda *_: None)
    random.seed(1234)
    got_train = t.run_epoch(0, "train")
    got_val = t.run_epoch(0, "val")

    ref = oracle.trainer.ReferenceTrainer(sd0, tc, arch="x3d_s", num_classes=3, abstain_class=False)
    random.seed(1234)  # the same affine configurations, drawn in the same order (one per micro-batch, train then val)
    want_train = ref.run_epoch(train, "train")
    want_val = ref.run_epoch(val, "val")
    assert t.current_iteration == ref.iteration == 4

    # parameters: compare the UPDATES (p - p0), each against its own scale; SGD: <= 1e-3 of the update's largest element
    sd1 = m.state_dict()
    worst = []
    for k, p0 in sd0.items():
        if not p0.is_floating_point() or k == "ones":
            continue
        if "running_" in k:
            scale = float(ref.sd[k].abs().max()) + 1e-12
            err = float((sd1[k].cpu() - ref.sd[k]).abs().max()) / scale
            assert err < 1e-4, f"{k}: running statistic differs by {err:.2e}"
            continue
        du_ref, du = ref.sd[k].detach() - p0, sd1[k].cpu() - p0
        scale = float(du_ref.abs().max())
        assert scale > 0, f"{k} did not move in the oracle"
        # what fp32 storage of the parameter itself allows: each of the two steps rounds p (magnitude max|p0|) on both sides
        floor = 8 * 2.0 ** -24 * float(p0.abs().max()) / scale
        worst.append((float((du - du_ref).abs().max()) / scale - floor, k, floor))
    worst.sort(reverse=True)
    if opt == "SGD":
        assert worst[0][0] < 1e-3, "largest relative update errors beyond the fp32 storage floor: " + ", ".join(f"{k} {e:.2e} (floor {f:.1e})" for e, k, f in worst[:6])
    else:  # Adam divides by sqrt(v): an element whose gradient is ~0 gets a full-size update of rounding-noise sign; judge the bulk
        bad = [w for w in worst if w[0] > 2e-2]
        assert len(bad) <= len(worst) // 20, "Adam updates differ on: " + ", ".join(f"{k} {e:.2e}" for e, k, _ in bad[:8])
    # the seven loss terms, summed over the epoch, training and validation (validation: eval-mode forward + eval-mode transform term)
    for got, want, tag in ((got_train, want_train, "train"), (got_val, want_val, "val")):
        nb = 4 if tag == "train" else 2
        terms = torch.tensor(got["loss_terms"], dtype=torch.float64) * nb
        for j, name in enumerate(oracle.trainer.TERM_NAMES):
            w = float(want["loss_terms_sum"][j])
            assert abs(float(terms[j]) - w) <= 2e-3 * max(abs(w), 1e-3), f"{tag} {name}: {float(terms[j]):.6g} vs oracle {w:.6g}"
        assert float(want["loss_terms_sum"][5]) != 0.0, "the transform term must be live in both modes"
    # same predictions -> same confusion matrix -> same scheduler / best-model decisions
    K = 3
    for got, want in ((got_train, want_train), (got_val, want_val)):
        cm = want["confusion"].double()
        tp, support, predicted = cm.diag(), cm.sum(1), cm.sum(0)
        f1 = torch.where(support + predicted > 0, 2 * tp / (support + predicted).clamp(min=1), torch.zeros_like(tp))
        assert got["f1"] == pytest.approx(f1.tolist(), abs=1e-12)


@pytest.mark.timeout(600)
def test_validation_transform_term_batched_equals_two_passes():
    """Eval epochs run [clips, warped clips] as ONE 2N-clip forward: identical (to fp32 rounding) to the reference's order of two N-clip
    passes, because eval-mode norm layers use running statistics."""
    import random

    from protoasnet_amd import losses

    m = synth_model(CFG_VIDEO_X3D).to(DEV).eval()
    x = _loader(1, 77, B=3)[0]["cine"].to(DEV)
    tl = losses.TransformLoss(loss_weight=1.0, reduction="mean")
    cfg = {"angle": 13.0, "scale": 1.2}
    with torch.no_grad():
        _, _, occ = m(x)
        two_pass = tl.compute(x, occ, m, config=cfg)
        both = m(torch.cat([x, losses.affine_warp(x, cfg["angle"], cfg["scale"])]))[2]
        one_pass = tl.compute_from_maps(both[:3], both[3:], cfg)
    assert float(two_pass) > 0 and abs(float(one_pass) - float(two_pass)) <= 1e-5 * float(two_pass)
