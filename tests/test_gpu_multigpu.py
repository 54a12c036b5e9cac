"""GPU, >= 2 cards: the multi-GPU path validates ITSELF on the first box that has more than one GPU (skipped on a 1-GPU box; the
world-2 gloo tests in test_cpu_distributed.py / test_gpu_dp.py cover the same host logic there).  Every rank is a fresh child process
started before it touches the GPU -- never an exec after GPU initialisation.

(a) ``bench.py --gpus 2`` over RCCL: both ranks join the collective, the line says so;
(b) ``pasn_allreduce`` through ``dp.NativeComm`` at world 2 equals ``torch.distributed.all_reduce`` on the same buffers, bit for bit;
(c) the data-parallel training step: the all-reduced gradients equal the mean of the ranks' local gradients, are identical on every rank,
    and equal what ONE process computes for the same global batch -- each rank's clips as its own pass (per-rank batch statistics, the
    reference's plain BatchNorm: SURVEY section 4.5 "BN frozen or per-rank to make that exact"), averaged.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _need_two_gpus():
    if torch.cuda.device_count() < 2:  # (counting devices does not initialise the GPU)
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one card)")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PASN_BENCH_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(900)
def test_bench_two_gpus_over_rccl():
    _need_two_gpus()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
                        "--frames", "4", "--size", "64", "--cpu-clips", "0", "--no-secondary"], env=_clean_env(), capture_output=True, text=True,
                       timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["collective_backend"] == "rccl"
    assert line["config"]["global_batch"] == 8 and line["scaling"] == "weak" and line["value"] > 0


@pytest.mark.timeout(900)
def test_native_allreduce_and_dp_gradients_world2(tmp_path):
    _need_two_gpus()
    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(_clean_env(), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "multigpu_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=840)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    outs = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    # (b)
    assert all(o["native_vs_torch"] == 0.0 for o in outs), [o["native_vs_torch"] for o in outs]
    # (c) exchange == mean of the local gradients, identical everywhere, in place on the flat buffer
    names = sorted(outs[0]["local"])
    assert names == sorted(outs[1]["local"]) and outs[0]["nbytes"] == outs[1]["nbytes"] > 0
    assert all(o["in_place"] for o in outs)
    for n in names:
        mean = (outs[0]["local"][n] + outs[1]["local"][n]) / 2
        for r in range(world):
            assert torch.allclose(outs[r]["reduced"][n], mean, rtol=1e-6, atol=1e-7 * float(mean.abs().max() + 1e-30)), (n, r)
        assert torch.equal(outs[0]["reduced"][n], outs[1]["reduced"][n]), n
    # ... and what ONE process computes for the same global batch (each rank's clips as its own pass, then the average)
    sys.path.insert(0, HERE)
    import multigpu_worker as W
    from util import CFG_VIDEO_X3D, synth_model

    dev = torch.device("cuda", 0)
    single = []
    for rank in range(world):
        model = synth_model(CFG_VIDEO_X3D).to(dev).train()
        W.train_step(model, W.clips(rank).to(dev), seed=rank)
        single.append({n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None})
    for n in names:
        want = (single[0][n] + single[1][n]) / 2
        scale = float(want.abs().max()) + 1e-30
        assert float((outs[0]["reduced"][n] - want).abs().max()) <= 1e-5 * scale, n
