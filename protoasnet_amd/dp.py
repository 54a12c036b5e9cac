"""Data-parallel gradient exchange: ONE flat fp32 bucket per optimizer step, all-reduced over RCCL / xGMI.

The reference trains on one GPU (no DistributedDataParallel anywhere in src/agents); BASELINE.json's config 3 asks for the
8-GPU data-parallel step.  Clips shard across ranks (one process per GPU), every rank runs the compiled forward + backward
launch lists on its micro-batch, and the only exchange is this all-reduce of the parameter gradients (X3D-S + head B:
about 3.8 M floats = 15 MB, one bucket: on the xGMI full mesh a ring all-reduce of M bytes moves 2*(n-1)/n*M per link
direction, ~0.2 ms at 8 GPUs -- far below the step, so it is not overlapped with the backward).  The compiled backward pass
already lands every gradient in one flat fp32 buffer (train.TrainPlan.backward), so the all-reduce runs IN PLACE on that buffer
(``flat_gradient_view``): no flatten / unflatten / copy-back passes around the one collective this workload has.
Norm-layer running statistics stay per rank, as they would under the reference's plain BatchNorm (no SyncBN).
"""
from __future__ import annotations

import ctypes
import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors

from . import _lib


class NativeComm:
    """RCCL communicator driven through the C-ABI (``pasn_comm_*`` / ``pasn_allreduce``): the native call site of the gradient exchange.

    The 128-byte RCCL id is made by rank 0 and reaches the other ranks over the EXISTING ``torch.distributed`` group (any backend: it is
    a one-off side channel, not the data path); after that ``all_reduce_`` is one ``ncclAllReduce`` on torch's current HIP stream, in
    place on the flat bucket.  One communicator per process (one process per GPU).  ``world_size == 1`` works without a process group.
    """

    _instance: Optional["NativeComm"] = None

    def __init__(self, rank: int, world_size: int, device: torch.device):
        lib = _lib.lib()
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (ctypes.c_ubyte * 128)()
            _lib.check(lib.pasn_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if world_size > 1:
            on_device = dist.get_backend() == "nccl"
            ident = ident.to(device) if on_device else ident
            dist.broadcast(ident, src=0)
            ident = ident.cpu()
        raw = (ctypes.c_ubyte * 128)(*ident.tolist())
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.pasn_comm_init(ctypes.cast(raw, ctypes.c_void_p), world_size, rank, ctypes.byref(handle)))
        self.handle, self.rank, self.world_size, self.device = handle, rank, world_size, device

    @classmethod
    def get(cls, device: torch.device) -> "NativeComm":
        if cls._instance is None:
            ws = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
            rk = dist.get_rank() if ws > 1 else 0
            cls._instance = cls(rk, ws, device)
        return cls._instance

    def all_reduce_(self, flat: torch.Tensor) -> torch.Tensor:
        assert flat.is_cuda and flat.is_contiguous() and flat.dtype in (torch.float32, torch.bfloat16)
        _lib.check(_lib.lib().pasn_allreduce(self.handle, flat.data_ptr(), flat.numel(), _lib.dtype_code(flat.dtype), _lib.current_stream()))
        return flat

    def close(self) -> None:
        if self.handle:
            _lib.check(_lib.lib().pasn_comm_destroy(self.handle))
            self.handle = ctypes.c_void_p()
        if NativeComm._instance is self:
            NativeComm._instance = None


def flat_gradient_view(grads) -> Optional[torch.Tensor]:
    """The ONE flat tensor the gradients already live in, or None.

    ``train.TrainPlan.backward`` writes every parameter gradient of a pass into one zero-initialised fp32 buffer (64-float aligned slots)
    and returns views of it; autograd's AccumulateGrad keeps those views as ``p.grad`` (later micro-batches accumulate into them in
    place).  When every gradient is a dense view of one storage, the span from the first to the last is the bucket as it lies: the
    all-reduce runs on it in place -- no flatten, no unflatten, no copy back.  (The alignment gaps inside the span are zeros on every
    rank and stay zeros under a sum.)  Anything else -- gradients from torch autograd, mixed dtypes, a span more than 25 % larger than
    the gradients it holds -- returns None and takes the bucketed path."""
    if not grads:
        return None
    g0 = grads[0]
    base = g0.untyped_storage().data_ptr()
    lo, hi, total = None, 0, 0
    for g in grads:
        if g.dtype != g0.dtype or g.device != g0.device or not g.is_contiguous() or g.untyped_storage().data_ptr() != base:
            return None
        o = g.storage_offset()
        lo = o if lo is None else min(lo, o)
        hi = max(hi, o + g.numel())
        total += g.numel()
    if hi - lo > total + total // 4 + 64 * len(grads):
        return None
    return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), lo, (hi - lo,))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None, average: bool = True,
                        native: Optional[bool] = None) -> int:
    """Sum (or average) ``p.grad`` of every parameter that has one across the ranks of ``group``; returns the bucket size in
    bytes (0 when not running distributed).  Every rank must hold gradients for the same parameters (they run the same graph).

    In-place path (``flat_gradient_view``): the exchange covers the SPAN of the arena's gradient buffer from the first to the last
    listed gradient -- the alignment gaps (zeros on every rank) and any gradient slot of a parameter NOT listed in ``params`` that lies
    inside the span are summed (and divided, with ``average``) too.  ``DPTrainer`` always passes every parameter of the model, so there
    is no such slot; a caller that passes a subset and must keep the rest local should pass gradients that are not arena views.

    ``native=True`` (or ``PASN_NATIVE_RCCL=1``) sends the bucket through ``pasn_allreduce`` -- RCCL called from the C-ABI library on
    torch's current stream -- instead of ``torch.distributed.all_reduce``; it needs CUDA gradients and the default (world) group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    if native is None:
        native = os.environ.get("PASN_NATIVE_RCCL") == "1"
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    flat = flat_gradient_view(grads)  # the training arena's gradient buffer as it lies (in place), else a fresh bucket
    # Every rank must take the SAME path with the SAME element count (round 5, ADVICE): the in-place view has the arena's layout (hi - lo
    # elements), the bucket the parameters' (total elements); if AccumulateGrad cloned a gradient on one rank, or one rank's p.grad
    # storages are mixed, the ranks would enter all_reduce with different sizes -- a hang or silent corruption.  One 3-word MIN-reduce
    # per call (a few microseconds next to the bucket itself): in place only when every rank can and the spans agree.
    total = sum(g.numel() for g in grads)
    span = flat.numel() if flat is not None else -1
    probe = torch.tensor([int(flat is not None), span, -span, total, -total], dtype=torch.int64, device=grads[0].device)
    if probe.is_cuda and dist.get_backend(group) == "gloo":
        probe = probe.cpu()
    dist.all_reduce(probe, op=dist.ReduceOp.MIN, group=group)
    if int(probe[3]) != -int(probe[4]):
        raise RuntimeError(f"allreduce_gradients: the ranks hold between {int(probe[3])} and {-int(probe[4])} gradient elements; every rank must "
                           "hold gradients for the same parameters")
    in_place = flat is not None and int(probe[0]) == 1 and int(probe[1]) == -int(probe[2])
    if not in_place:
        flat = _flatten_dense_tensors(grads)
    if native and flat.is_cuda and group is None:
        NativeComm.get(flat.device).all_reduce_(flat)
    elif flat.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (several ranks sharing one GPU, where RCCL refuses to start): stage the bucket through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)  # RCCL over xGMI on a real node
    if average:
        flat /= dist.get_world_size(group)
    if not in_place:
        for g, f in zip(grads, _unflatten_dense_tensors(flat, grads)):
            g.copy_(f)
    return flat.numel() * flat.element_size()
