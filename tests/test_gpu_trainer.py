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
