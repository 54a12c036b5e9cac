"""hipGraph replay of an eval-mode forward (the whole launch list -- trunk plan + head -- as ONE graph launch).

The launch lists of this package are static per (input shape, dtype, weights): 72 trunk + 2 head launches for X3D-S.  Captured once
on a side stream (``torch.cuda.CUDAGraph``; the C-ABI launches go to torch's current stream, which is the capturing stream during
capture) and replayed, the host enqueues one packet per step instead of ~75 and the device sees the kernels back to back:
2.78 -> 2.74 ms per 32-clip batch on MI355X (profiles/README.md entry 106).

The graph is tied to the INPUT TENSOR'S STORAGE (its address is a kernel argument): call it with the same tensor object's storage and
refresh the clip with ``x.copy_(new_clip)``; a tensor at another address, shape or dtype gets its own capture.  The returned tensors
are the graph's static outputs: every replay overwrites them (clone what must survive the next call).
"""
from typing import Dict, Tuple

import torch


class GraphedForward:
    def __init__(self, model: torch.nn.Module, warmup: int = 3):
        if model.training:
            raise RuntimeError("GraphedForward replays the eval-mode forward: call model.eval() first")
        self.model, self.warmup = model, max(1, int(warmup))
        self._graphs: Dict[Tuple, Tuple[torch.cuda.CUDAGraph, object]] = {}

    def _capture(self, x: torch.Tensor):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # compiles / packs the plan and settles every lazily created buffer before the capture
                self.model(x)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        # thread_local: calls other threads make while this one captures (the RCCL watchdog of an initialised process group polls events) do not
        # invalidate the capture
        with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = self.model(x)
        return g, out

    def __call__(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("GraphedForward runs on the GPU only; there is no CPU fallback")
        key = (x.data_ptr(), tuple(x.shape), x.dtype, tuple(x.stride()))
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = self._capture(x)
        ent[0].replay()
        return ent[1]

    def clear(self) -> None:
        """Drop the captured graphs (after a weight update: the packed weights and folded norms inside them are stale)."""
        self._graphs.clear()
