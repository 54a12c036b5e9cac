"""Rank process of tests/test_gpu_multigpu.py (one process per GPU, RCCL between them).  Started as a fresh interpreter BEFORE it touches
the GPU: ``python multigpu_worker.py <out_dir>`` with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment.

Writes ``<out_dir>/r<rank>.pt``:
  native_vs_torch  max |pasn_allreduce(NativeComm) - torch.distributed.all_reduce| on the same fp32 / bf16 buffers (must be 0)
  local / reduced  this rank's parameter gradients of one training step on ITS clips, before / after the gradient exchange
  in_place         whether the exchange ran on the training pass's flat gradient buffer as it lies
"""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def clips(rank):
    from protoasnet_amd import synth

    return synth.echo_clips((2, 3, 4, 64, 64), seed=100 + rank)


def train_step(model, x, seed):
    g = torch.Generator().manual_seed(seed)
    logits, sim, occ = model(x)
    w = torch.randn(logits.shape, generator=g).to(x.device)
    ((logits * w).sum() + sim.sum() + 0.1 * occ.sum()).backward()


def main(out_dir):
    from protoasnet_amd import dp
    from util import CFG_VIDEO_X3D, synth_model

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    res = {}
    # (b) the native call site against torch.distributed on the same data
    comm = dp.NativeComm.get(dev)
    assert comm.world_size == world and comm.rank == rank
    g = torch.Generator().manual_seed(7 + rank)
    worst = 0.0
    for dtype, n in ((torch.float32, 3_800_000), (torch.bfloat16, 1 << 16), (torch.float32, 1)):
        a = torch.randn(n, generator=g).to(dev).to(dtype)
        b = a.clone()
        comm.all_reduce_(a)
        dist.all_reduce(b)
        torch.cuda.synchronize()
        worst = max(worst, float((a.float() - b.float()).abs().max()))
    res["native_vs_torch"] = worst
    # (c) one data-parallel training step: compiled forward + backward on this rank's clips, ONE exchange of the flat gradient buffer
    model = synth_model(CFG_VIDEO_X3D).to(dev).train()
    train_step(model, clips(rank).to(dev), seed=rank)
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    res["in_place"] = dp.flat_gradient_view(grads) is not None
    res["local"] = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
    res["nbytes"] = dp.allreduce_gradients(model.parameters(), native=True)
    torch.cuda.synchronize()
    res["reduced"] = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
