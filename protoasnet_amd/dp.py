"""Data-parallel gradient exchange: ONE flat fp32 bucket per optimizer step, all-reduced over RCCL / xGMI.

The reference trains on one GPU (no DistributedDataParallel anywhere in src/agents); BASELINE.json's config 3 asks for the
8-GPU data-parallel step.  Clips shard across ranks (one process per GPU), every rank runs the compiled forward + backward
launch lists on its micro-batch, and the only exchange is this all-reduce of the parameter gradients (X3D-S + head B:
about 3.8 M floats = 15 MB, one bucket: on the xGMI full mesh a ring all-reduce of M bytes moves 2*(n-1)/n*M per link
direction, ~0.2 ms at 8 GPUs -- far below the step, so it is not overlapped with the backward in this round).
Norm-layer running statistics stay per rank, as they would under the reference's plain BatchNorm (no SyncBN).
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None, average: bool = True) -> int:
    """Sum (or average) ``p.grad`` of every parameter that has one across the ranks of ``group``; returns the bucket size in
    bytes (0 when not running distributed).  Every rank must hold gradients for the same parameters (they run the same graph)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    flat = _flatten_dense_tensors(grads)
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (several ranks sharing one GPU, where RCCL refuses to start): stage the bucket through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)  # RCCL over xGMI on a real node
    if average:
        flat /= dist.get_world_size(group)
    for g, f in zip(grads, _unflatten_dense_tensors(flat, grads)):
        g.copy_(f)
    return flat.numel() * flat.element_size()
