"""hipGraph replay of an eval-mode forward (the whole launch list -- trunk plan + head -- as ONE graph launch).

The launch lists of this package are static per (input shape, dtype, weights): 72 trunk + 2 head launches for X3D-S.  Captured once
on a side stream (``torch.cuda.CUDAGraph``; the C-ABI launches go to torch's current stream, which is the capturing stream during
capture) and replayed, the host enqueues one packet per step instead of ~75 and the device sees the kernels back to back:
2.78 -> 2.74 ms per 32-clip batch on MI355X (profiles/README.md entry 106).

What a captured graph depends on, and how each dependency is tracked:

* **the input's address** -- a kernel argument.  Default (``static_input=False``): zero-copy, one capture per (storage address, shape, dtype,
  strides); refresh the clip with ``x.copy_(new_clip)``.  At most ``max_graphs`` captures are kept (least recently used evicted, with a
  warning: a caller that hands over a fresh tensor every step is re-capturing every step and should use ``static_input=True``).
  ``static_input=True``: the wrapper owns one input buffer per (shape, dtype) and copies the caller's tensor into it before the replay
  (one extra pass over the clip, ~2 % of a step at 32x3x16x224x224 bf16) -- any tensor may be passed.
* **the weights** -- the trunk plan's packed weights / folded norms and the head's packed 1x1 convs are baked into the graph as
  addresses of tensors created during the warm-up.  Every call compares the (storage address, version counter) of EVERY parameter and
  buffer of the model with the ones seen at capture: after ``load_state_dict``, an optimizer step, ``DPTrainer.sync_*`` or any other
  in-place update the stale graph is dropped and the forward is captured again -- it is never replayed.  ``prototype_vectors`` and
  ``last_layer.weight`` are read in place by the head kernels (never packed), so the reference's ``.data.copy_()`` write of a push
  (push_abs_revision.py:346), which bumps no counter, is seen by the next replay as it is.  A ``.data`` write to any OTHER tensor must
  be followed by ``clear()`` (and ``trunk.invalidate_plans()``), exactly as for the eager path (plan.HipTrunk.invalidate_plans).
* **the memory behind those addresses** -- each cache entry holds references to the trunk's ``Plan`` objects (arena, packed weights)
  and the head chains' packed tensors it was captured with, so an eager ``model(x)`` that re-plans after a weight change cannot hand
  the arena back to the allocator while a graph that points into it is still alive.

The returned tensors are the graph's static outputs: every replay overwrites them (clone what must survive the next call).
"""
import warnings
from collections import OrderedDict
from typing import Tuple

import torch

from . import _lib


class _Entry:
    __slots__ = ("graph", "out", "sig", "keep", "x_static")

    def __init__(self, graph, out, sig, keep, x_static):
        self.graph, self.out, self.sig, self.keep, self.x_static = graph, out, sig, keep, x_static


class GraphedForward:
    def __init__(self, model: torch.nn.Module, warmup: int = 3, static_input: bool = False, max_graphs: int = 4):
        if model.training:
            raise RuntimeError("GraphedForward replays the eval-mode forward: call model.eval() first")
        self.model, self.warmup = model, max(1, int(warmup))
        self.static_input, self.max_graphs = bool(static_input), max(1, int(max_graphs))
        self._graphs: "OrderedDict[Tuple, _Entry]" = OrderedDict()
        self.captures = 0  # how many times a forward was captured (tests; a steadily growing count means the key keeps changing)

    # ------------------------------------------------------------------------------------------ what the graph was captured against
    def _signature(self) -> tuple:
        return tuple((t.data_ptr(), t._version) for t in list(self.model.parameters()) + list(self.model.buffers()))

    def _keepalive(self) -> list:
        """The objects whose device memory the captured kernels address but the graph's private pool does not own."""
        keep = []
        for m in self.model.modules():
            plans = getattr(m, "_plans", None)  # plan.HipTrunk: {key: (signature, Plan)}
            if isinstance(plans, dict):
                keep.append(dict(plans))
            cache = getattr(m, "_cache", None)  # nets.PointwiseChain: packed weights / row plans
            if isinstance(cache, dict):
                keep.append(dict(cache))
        return keep

    def _capture(self, x: torch.Tensor, sig: tuple, x_static) -> _Entry:
        if self.model.training:
            raise RuntimeError("GraphedForward replays the eval-mode forward: the model was switched to train() since construction")
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
        self.captures += 1
        return _Entry(g, out, sig, self._keepalive(), x_static)

    def __call__(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("GraphedForward runs on the GPU only; there is no CPU fallback")
        if self.static_input:
            key = ("static", tuple(x.shape), x.dtype, x.device, _lib.tuning_epoch())
        else:
            key = (x.data_ptr(), tuple(x.shape), x.dtype, tuple(x.stride()), x.device, _lib.tuning_epoch())
        sig = self._signature()
        ent = self._graphs.get(key)
        if ent is not None and ent.sig != sig:  # a weight / buffer changed since the capture: never replay the stale graph
            del self._graphs[key]
            ent = None
        if ent is None:
            x_static = None
            if self.static_input:
                x_static = torch.empty_like(x, memory_format=torch.contiguous_format)
                x_static.copy_(x)
            ent = self._capture(x_static if self.static_input else x, sig, x_static)
            self._graphs[key] = ent
            while len(self._graphs) > self.max_graphs:
                old, _ = self._graphs.popitem(last=False)
                warnings.warn(
                    f"GraphedForward: more than {self.max_graphs} live captures -- evicting the one for input {old[:3]}.  A new input address "
                    "per call re-captures per call; pass static_input=True (or reuse one input tensor and copy_ into it)", RuntimeWarning)
        else:
            self._graphs.move_to_end(key)
            if ent.x_static is not None:
                ent.x_static.copy_(x)
        ent.graph.replay()
        return ent.out

    def clear(self) -> None:
        """Drop the captured graphs (needed only after a write that bumps no version counter, e.g. through ``param.data`` of a packed tensor)."""
        self._graphs.clear()
