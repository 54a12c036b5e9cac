"""Training path: train-mode forward (batch-statistics norm) and backward as ONE replayable list of HIP launches.

The reference trains by ``loss.backward()`` through ``model(x)`` / ``model.compute_occurence_map(x)``
(src/agents/Video_XProtoNet_e2e.py:118-141, src/loss/loss.py:302).  Here the whole model (trunk + head) is compiled, for
one input shape, into a forward launch list and a backward launch list by a small tape: every conv + norm + activation
*unit* emits its forward launches immediately and registers an emitter for its backward launches, which are generated in
reverse order once the forward graph is complete.  Both lists share one arena whose offsets come from the same live-range
analysis as the inference plan -- an activation stays resident exactly until the last backward launch that reads it.

Parameters stay fp32 ``nn.Parameter``s (the optimizer and the RCCL gradient all-reduce see ordinary ``.grad`` tensors);
activations and activation gradients are fp32 or bf16 (``set_compute_dtype``); statistics, reductions and parameter
gradients are fp32.  ``TrainPlan`` is driven by ``_TrainFn`` (a ``torch.autograd.Function``), so the reference's losses
and optimizers sit on top unchanged.  Kernels: ``csrc/train.hip``, ``csrc/wgrad.hip``, ``csrc/head_train.hip`` plus the
inference conv kernels (a 1x1x1 conv's input gradient is the same kernel with the transposed weight).
"""
from __future__ import annotations

import ctypes
import os
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import ConvDesc, XProtoDesc
from .plan import ALIGN, Act, PlanBuilder, _triple, round_up


def _need_fp32_param(p: torch.Tensor, what: str) -> None:
    if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
        raise RuntimeError(
            f"training keeps fp32 master parameters on the GPU ({what} is {p.dtype} on {p.device}); "
            "use model.set_compute_dtype(torch.bfloat16) for bf16 activations instead of casting the model"
        )


class TrainBuilder(PlanBuilder):
    def __init__(self, device, dtype, in_dtype, groups: int = 1):
        super().__init__(device, dtype, in_dtype)
        # statistics groups: the batch is `groups` runs of N / groups clips, each normalised with its own batch statistics (2: the clips and
        # their warped copies of the reference's loss recipe in ONE pass -- TrainRunner mode 2)
        self.groups = int(groups)
        self.tape: List[Callable[[], None]] = []
        self.grads: Dict[int, Act] = {}        # activation buffer id -> Act holding its gradient
        self.readers: Dict[int, int] = {}      # activation buffer id -> number of ops that consume it (conv input, residual, pool, head)
        self.red_hook: Dict[int, dict] = {}    # activation buffer id -> how its producer unit's backward sums can be taken by a fused dgrad
        self.refresh: List[Callable[[], None]] = []
        # weights packed by the native one-launch packer (pasn_pack_weights): (parameter, destination, mode, cout, cin, taps, rows, kc,
        # frag, kstep, ch) -- source dims as the PARAMETER has them.  PASN_NO_PACK=1: torch expressions per parameter (the old path)
        self.pack_jobs: List[tuple] = []
        self.native_pack = _lib.tuning_get("PASN_NO_PACK") != "1"
        self.pslots: List[Tuple[torch.Tensor, int, int]] = []
        self._slot_of: Dict[int, int] = {}
        self.gsize = 0
        self.gbuf = self._new_buf(0, external=True)
        self._const: Dict[tuple, torch.Tensor] = {}
        self.nbt: List[torch.Tensor] = []      # num_batches_tracked counters bumped once per forward
        self.n_fwd = -1
        self.op_names: List[str] = []
        self.op_bytes: List[int] = []          # bytes of the arena / external buffers a launch touches, each buffer once (bench roofline)
        self._pending: Dict[int, int] = {}
        # Weight-gradient launches have no reader before the optimizer: they go to a SECOND stream and overlap the main chain (the next
        # units' reduce / apply / input-gradient passes).  op_kind: 0 main stream, 1 side stream, 2 join (the main stream waits for the
        # side launch op_join names; the buffers that launch touches stay live in the arena until then).  At most ``side_depth`` side
        # launches are outstanding.  PASN_TRAIN_STREAMS=1: everything on one stream.
        self.op_kind: List[int] = []
        self.op_join: Dict[int, int] = {}
        self._forks: List[Tuple[int, tuple]] = []
        self.side_depth = 0 if _lib.tuning_get("PASN_TRAIN_STREAMS") == "1" else int(_lib.tuning_get("PASN_TRAIN_SIDE_DEPTH") or 2)
        # activation buffer id -> does anything that produced it hold a parameter with requires_grad?  (The reference's agents
        # freeze the trunk / everything but the last layer in some phases: XProtoNet_Base.py:253-293; frozen parts get no
        # backward launches at all.)
        self.live: Dict[int, bool] = {}

    # ---- small helpers -------------------------------------------------------------------------------------------
    def const(self, rows: int, value: float) -> torch.Tensor:
        key = (rows, value)
        if key not in self._const:
            self._const[key] = torch.full((rows,), value, dtype=torch.float32, device=self.device)
        return self._const[key]

    def slot(self, p: torch.Tensor) -> int:
        """Offset (in floats) of ``p``'s gradient inside the flat fp32 gradient buffer of one backward pass."""
        if id(p) not in self._slot_of:
            _need_fp32_param(p, "a parameter")
            self._slot_of[id(p)] = self.gsize
            self.pslots.append((p, self.gsize, p.numel()))
            self.gsize += round_up(p.numel(), 64)
        return self._slot_of[id(p)]

    def like(self, a: Act) -> Act:
        return Act(a.N, a.T, a.H, a.W, a.C, a.Cp, self._new_buf(a.N * a.positions * a.Cp * self.es))

    # ---- launch recording -----------------------------------------------------------------------------------------
    def _op(self, fn, *args) -> None:
        """Record one C-ABI call.  Each argument is a constant, or a callable of the per-run pointer table (``B`` / ``Pm`` /
        ``Gp`` below); everything is bound HERE, so later re-use of a local name cannot change a recorded launch."""
        bound = tuple(args)

        def run(ptrs, st, fn=fn, bound=bound):
            _lib.check(fn(*[a(ptrs) if callable(a) else a for a in bound], st))

        self.ops.append(run)
        self.op_kind.append(0)
        self.op_names.append(getattr(fn, "__name__", "?"))
        self.op_bytes.append(sum(self._pending.values()))
        self._pending = {}

    def _side_op(self, fn, bufs: tuple, *args) -> None:
        """A launch nothing downstream reads before the optimizer (a weight gradient): recorded for the side stream."""
        if self.side_depth <= 0:
            self._use(*bufs)
            self._op(fn, *args)
            return
        while len(self._forks) >= self.side_depth:
            self._join()
        self._use(*bufs)
        self._op(fn, *args)
        self.op_kind[-1] = 1
        self._forks.append((len(self.ops) - 1, bufs))

    def _join(self) -> None:
        fork, bufs = self._forks.pop(0)
        self._use(*bufs)  # the side launch's operands and scratch stay where they are until the main stream has waited for it
        self._pending = {}
        self.op_join[len(self.ops)] = fork
        self.ops.append(lambda ptrs, st: None)
        self.op_kind.append(2)
        self.op_names.append("join")
        self.op_bytes.append(0)

    def _use(self, *buf_ids) -> None:
        super()._use(*buf_ids)
        for b in buf_ids:
            if b is not None:
                self._pending[b] = self.bufs[b].nbytes

    @staticmethod
    def B(buf: Optional[int]):
        """Arena / external buffer id -> its address in this run (None -> NULL)."""
        return 0 if buf is None else (lambda ptrs, b=buf: ptrs[b])

    @staticmethod
    def Pm(t: Optional[torch.Tensor]):
        """Live fp32 parameter / buffer -> its address at launch time (the optimizer updates it in place)."""
        return 0 if t is None else (lambda ptrs, t=t: t.data_ptr())

    def Gp(self, off: int):
        """Address of a parameter-gradient slot inside this backward pass's flat gradient buffer."""
        return lambda ptrs, g=self.gbuf, o=4 * off: ptrs[g] + o

    def Gof(self, p: Optional[torch.Tensor], nullable: bool = True):
        """Where a kernel writes ``p``'s gradient: its slot, or -- for a frozen parameter -- NULL (``nullable``) / arena scratch."""
        if p is not None and p.requires_grad:
            return self.Gp(self.slot(p))
        if nullable or p is None:
            return 0
        scratch = self._new_buf(p.numel() * 4)
        self._use(scratch)
        return self.B(scratch)

    def add_grad(self, a: Act, g: Act) -> None:
        have = self.grads.get(a.buf)
        if have is None:
            self.grads[a.buf] = g
            return
        self._use(have.buf, g.buf)
        self._op(self.lib.pasn_add_inplace, self.B(have.buf), self.B(g.buf), a.N * a.positions * a.Cp, self.code)

    # ---- dense conv launch with a weight that is re-packed from the live parameter every step --------------------------
    def _dense(self, x: Act, y: Act, k, s, p, cout: int, cin: int, weight_fn: Callable[[], torch.Tensor],
               residual: Optional[Act] = None, pack: Optional[Tuple[torch.Tensor, int]] = None) -> ConvDesc:
        """``pack`` = (parameter [cout][cin][taps...] fp32, mode 0 forward / 1 input gradient): what ``weight_fn`` computes, for the native packer."""
        taps = k[0] * k[1] * k[2]
        kstep, ch = (16, 8) if self.dtype == torch.bfloat16 else (8, 4)
        kc, rows = round_up(x.Cp, kstep), round_up(round_up(cout, 8), 128)
        wp = torch.zeros(rows, taps, kc, dtype=self.dtype, device=self.device)
        d = self._desc(x, y, k, s, p, "none", False, kc, rows)
        dref = ctypes.byref(d)
        frag = 2500 <= int(self.lib.pasn_conv3d_variant(dref, self.code, 0)) < 6000  # x-tile kernels: fragment-major weights; 6000+: implicit GEMM, plain
        wf = torch.empty_like(wp) if frag else None
        if frag:
            d.w_frag = 1

        def refresh():
            wp[:cout, :, :cin] = weight_fn().reshape(cout, cin, taps).permute(0, 2, 1)
            if frag:
                wf.view(rows // 32, kc // kstep, 2, 32, ch).copy_(wp.view(rows // 32, 32, kc // kstep, 2, ch).permute(0, 2, 3, 1, 4))

        if pack is not None and self.native_pack and pack[0].is_contiguous() and (not frag or taps == 1):
            param, mode = pack
            src_cout, src_cin = (cout, cin) if mode == 0 else (cin, cout)
            assert param.numel() == src_cout * src_cin * taps
            self.pack_jobs.append((param, wf if frag else wp, mode, src_cout, src_cin, taps, rows, kc, int(frag), kstep, ch))
        else:
            self.refresh.append(refresh)
        one, zero = self.const(rows, 1.0), self.const(rows, 0.0)
        self.keep += [wp, wf, one, zero]
        rb = residual.buf if residual is not None else None
        self._use(x.buf, y.buf, rb)
        self._op(self.lib.pasn_conv3d_fwd, self.B(x.buf), (wf if frag else wp).data_ptr(), one.data_ptr(), zero.data_ptr(), self.B(rb), 0,
                 self.B(y.buf), dref, self.code)
        return d

    # ---- one conv (+ norm) (+ squeeze-excite) (+ residual) + activation unit ---------------------------------------------
    def unit(self, x: Act, conv: nn.Module, norm: Optional[nn.Module], act: str, kind: str = "conv",
             residual: Optional[Act] = None, se: Optional[nn.Module] = None) -> Act:
        k, s, p = _triple(conv.kernel_size, 1), _triple(conv.stride, 1), _triple(conv.padding, 0)
        cout, cin = conv.out_channels, conv.in_channels
        _need_fp32_param(conv.weight, "a conv weight")
        taps = k[0] * k[1] * k[2]
        y = self._out_act(x, cout, k, s, p)
        N, S, C, Cp = y.N, y.positions, y.C, y.Cp
        code, lib, B, Pm = self.code, self.lib, self.B, self.Pm
        actc = _lib.ACT[act]
        dw_w = None
        self.readers[x.buf] = self.readers.get(x.buf, 0) + 1
        if residual is not None:
            self.readers[residual.buf] = self.readers.get(residual.buf, 0) + 1
        # ---------------- forward conv (raw output) ----------------
        if kind == "first":
            assert x.planar and x.C == 3 and k[0] == 1 and s[0] == 1 and p[0] == 0
            d = self._desc(x, y, k, s, p, "none")
            one, zero = self.const(Cp, 1.0), self.const(Cp, 0.0)
            slot = int(lib.pasn_first_conv_mfma_slot(ctypes.byref(d), _lib.dtype_code(self.in_dtype), code))
            self._use(x.buf, y.buf)
            if slot >= 0:
                # the 7x7 stride-2 stems with bf16 activations: matrix-core kernel, weights in its K order (rows (ci, r), 8-wide window slots)
                rows, nq, bn = 3 * k[1], 2 * ((3 * k[1] + 1) // 2), 32 * ((Cp + 31) // 32)
                wq32 = torch.zeros(nq, bn, 8, dtype=torch.float32, device=self.device)
                wq = torch.zeros(nq, bn, 8, dtype=torch.bfloat16, device=self.device)

                def refresh_first():
                    wq32[:rows, :C, slot:slot + k[2]] = conv.weight.detach().reshape(C, 3, k[1], k[2]).permute(1, 2, 0, 3).reshape(rows, C, k[2])
                    wq.copy_(wq32)

                self.refresh.append(refresh_first)
                self.keep += [wq32, wq, one, zero]
                self._op(lib.pasn_first_conv_mfma_fwd, B(x.buf), wq.data_ptr(), one.data_ptr(), zero.data_ptr(), B(y.buf), ctypes.byref(d),
                         _lib.dtype_code(self.in_dtype), 1.0, 0.0)
            else:
                wfirst = torch.zeros(3 * k[1] * k[2], Cp, dtype=torch.float32, device=self.device)
                self.refresh.append(lambda: wfirst[:, :C].copy_(conv.weight.detach().reshape(C, 3, k[1], k[2]).permute(1, 2, 3, 0).reshape(3 * k[1] * k[2], C)))
                self.keep += [wfirst, one, zero]
                self._op(lib.pasn_first_conv_fwd, B(x.buf), wfirst.data_ptr(), one.data_ptr(), zero.data_ptr(), B(y.buf), ctypes.byref(d),
                         _lib.dtype_code(self.in_dtype), code)
        elif kind == "dw":
            assert conv.groups == cin == cout == x.C and not x.planar
            dw_w = torch.zeros(taps, Cp, dtype=torch.float32, device=self.device)
            d = self._desc(x, y, k, s, p, "none")
            if self.native_pack and conv.weight.is_contiguous():
                self.pack_jobs.append((conv.weight, dw_w, 2, C, 1, taps, taps, Cp, 0, 0, 0))
            else:
                self.refresh.append(lambda: dw_w[:, :C].copy_(conv.weight.detach().reshape(C, taps).t()))
            one, zero = self.const(Cp, 1.0), self.const(Cp, 0.0)
            self.keep += [dw_w, one, zero]
            # the stencil can take the unit's batch statistics in the same pass over y (saves the read of y by pasn_bn_stats_fwd)
            dw_rows = int(lib.pasn_dwconv3d_stats_rows(ctypes.byref(d), code)) if norm is not None and norm.momentum is not None else 0
            if dw_rows == 0:
                self._use(x.buf, y.buf)
                self._op(lib.pasn_dwconv3d_fwd, B(x.buf), dw_w.data_ptr(), one.data_ptr(), zero.data_ptr(), B(y.buf), 0, ctypes.byref(d), code)
        else:
            assert conv.groups == 1 and not x.planar
            d = self._dense(x, y, k, s, p, cout, cin, lambda: conv.weight.detach(), pack=(conv.weight, 0))
        dref = ctypes.byref(d)
        # ---------------- statistics / affine ----------------
        plain = norm is None and conv.bias is None and act == "none" and residual is None and se is None
        stat_buf = pool_buf = gate_buf = None
        stat = 0  # address source of the unit's (mean, invstd, sc, sh) table
        if norm is not None:
            assert conv.bias is None, "a conv followed by a norm layer carries no bias in the trunks built here"
            for t in (norm.weight, norm.bias):
                _need_fp32_param(t, "a norm parameter")
            if norm.momentum is None:
                raise NotImplementedError("cumulative-average BatchNorm (momentum=None) is not built")
            fused_dw = kind == "dw" and dw_rows > 0
            chunks = dw_rows if fused_dw else int(lib.pasn_train_chunks(N, S, Cp))
            ws = self._new_buf(N * chunks * 2 * Cp * 4)
            stat_buf = self._new_buf(self.groups * 4 * Cp * 4)
            pool_buf = self._new_buf(N * Cp * 4) if se is not None else None
            track = bool(norm.track_running_stats and norm.running_mean is not None)
            if track:
                self.nbt.append(norm.num_batches_tracked)
            if fused_dw:
                self._use(x.buf, y.buf, ws, stat_buf, pool_buf)
                self._op(lib.pasn_dwconv3d_stats_fwd_g, B(x.buf), dw_w.data_ptr(), one.data_ptr(), zero.data_ptr(), B(y.buf), B(ws), Pm(norm.weight),
                         Pm(norm.bias), Pm(norm.running_mean if track else None), Pm(norm.running_var if track else None), float(norm.momentum),
                         float(norm.eps), B(stat_buf), B(pool_buf), dref, code, self.groups)
            else:
                self._use(y.buf, ws, stat_buf, pool_buf)
                self._op(lib.pasn_bn_stats_fwd_g, B(y.buf), B(ws), Pm(norm.weight), Pm(norm.bias), Pm(norm.running_mean if track else None),
                         Pm(norm.running_var if track else None), float(norm.momentum), float(norm.eps), B(stat_buf), B(pool_buf), N, S, C, Cp, code,
                         self.groups)
            stat = B(stat_buf)
        elif not plain:
            stat_t = torch.zeros(self.groups, 4, Cp, dtype=torch.float32, device=self.device)  # per group (mean 0, invstd 1, sc 1, sh = bias)
            stat_t[:, 1, :C] = 1.0
            stat_t[:, 2, :C] = 1.0
            if conv.bias is not None:
                _need_fp32_param(conv.bias, "a conv bias")
                self.refresh.append(lambda: stat_t[:, 3, :C].copy_(conv.bias.detach().expand(self.groups, C)))
            self.keep.append(stat_t)
            stat = stat_t.data_ptr()
        if se is not None:
            assert norm is not None
            cse = se.fc1.out_channels
            for t in (se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias):
                _need_fp32_param(t, "a squeeze-excite parameter")
            gate_buf = self._new_buf(N * Cp * 4)
            self._use(pool_buf, gate_buf)
            self._op(lib.pasn_se_gate_fwd, B(pool_buf), 1, 1, Pm(se.fc1.weight), Pm(se.fc1.bias), Pm(se.fc2.weight), Pm(se.fc2.bias), B(gate_buf),
                     N, C, Cp, cse)
        if plain:
            out = y
        else:
            out = self.like(y)
            if residual is not None:
                assert (residual.N, residual.positions, residual.Cp) == (N, S, Cp)
            rb = residual.buf if residual is not None else None
            self._use(y.buf, out.buf, rb, stat_buf, gate_buf)
            self._op(lib.pasn_affine_act_fwd_g, B(y.buf), stat, B(rb), B(gate_buf), B(out.buf), N, S, C, Cp, actc, code, self.groups)

        if norm is not None and se is None and residual is None and not plain and self.groups == 1:
            # a consumer whose input gradient is a stencil launch (stride-1 depthwise dgrad) may take this unit's backward sums in that
            # launch: it leaves the coefficient buffer in hook["coef"], and backward() below then skips its own reduce pass
            self.red_hook[out.buf] = {"y_buf": y.buf, "stat": stat, "stat_buf": stat_buf, "actc": actc,
                                      "dg": lambda: self.Gof(norm.weight), "db": lambda: self.Gof(norm.bias), "coef": None}
        own = [conv.weight, conv.bias] + ([norm.weight, norm.bias] if norm is not None else []) + \
              ([se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias] if se is not None else [])
        own_live = any(t is not None and t.requires_grad for t in own)
        x_live = self.live.get(x.buf, False)
        res_live = residual is not None and self.live.get(residual.buf, False)
        self.live[out.buf] = x_live or own_live or res_live
        w_live = conv.weight.requires_grad

        # ---------------- backward emitter ----------------
        def backward() -> None:
            g = self.grads.get(out.buf)
            if g is None or not self.live[out.buf]:
                return  # nothing downstream needs this unit's gradient / nothing in or before it is trainable
            dy = g
            if not plain:
                chunks = int(lib.pasn_train_chunks(N, S, Cp))
                ws = self._new_buf(N * chunks * 2 * Cp * 4)
                coef = self._new_buf(self.groups * 2 * Cp * 4)
                rb = residual.buf if residual is not None else None
                red = lib.pasn_unit_bwd_reduce_g
                if norm is not None:
                    dg, db = self.Gof(norm.weight), self.Gof(norm.bias)
                elif conv.bias is not None:
                    dg, db = 0, self.Gof(conv.bias)
                else:
                    dg = db = 0
                se_analytic = False
                lazy = se is None and norm is not None and residual is None  # nobody but the apply pass reads the differentiated d
                hook = self.red_hook.get(out.buf)
                if hook is not None and hook["coef"] is not None:
                    coef = hook["coef"]  # the consumer's fused dgrad already took the sums (and dgamma / dbeta)
                elif se is None:
                    self._use(g.buf, y.buf, rb, stat_buf, ws, coef)
                    self._op(red, 3 if lazy else 0, B(g.buf), B(y.buf), stat, B(rb), 0, 0, B(ws), B(coef), dg, db, N, S, C, Cp, actc, code, self.groups)
                elif residual is None and not _lib.tuning_get("PASN_NO_SE_ANALYTIC"):
                    # squeeze-excite unit, ONE pass over (d, y): mode 4 leaves d' = d act'(.) and per-clip (sum d', sum d' yhat, sum yhat);
                    # the gate's gradient, the norm's coefficients and dgamma / dbeta follow from those per clip (d'' = d' gate + add is
                    # affine in d'), and the apply pass forms d'' on the fly -- the second pass over the tensor (mode 2) is gone
                    cse = se.fc1.out_channels
                    se_analytic = True
                    addb = self._new_buf(N * Cp * 4)
                    pn = self._new_buf(int(lib.pasn_se_bwd_workspace_floats(N, C, cse)) * 4)
                    ws3 = self._new_buf(N * chunks * 3 * Cp * 4)
                    o = [self.Gof(t, nullable=False) for t in (se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias)]
                    self._use(g.buf, y.buf, stat_buf, gate_buf, ws3)
                    self._op(red, 4, B(g.buf), B(y.buf), stat, 0, B(gate_buf), 0, B(ws3), 0, 0, 0, N, S, C, Cp, actc, code, self.groups)
                    self._use(ws3, pool_buf, stat_buf, gate_buf, addb, pn, coef)
                    self._op(lib.pasn_se_gate_bwd_stat_g, B(ws3), B(pool_buf), stat, B(gate_buf), Pm(se.fc1.weight), Pm(se.fc1.bias),
                             Pm(se.fc2.weight), Pm(se.fc2.bias), B(addb), B(pn), o[0], o[1], o[2], o[3], B(coef), dg, db, N, S, C, Cp, cse, self.groups)
                else:
                    cse = se.fc1.out_channels
                    addb = self._new_buf(N * Cp * 4)
                    pn = self._new_buf(int(lib.pasn_se_bwd_workspace_floats(N, C, cse)) * 4)
                    o = [self.Gof(t, nullable=False) for t in (se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias)]
                    self._use(g.buf, y.buf, stat_buf, gate_buf, ws)
                    self._op(red, 1, B(g.buf), B(y.buf), stat, 0, B(gate_buf), 0, B(ws), 0, 0, 0, N, S, C, Cp, actc, code, self.groups)
                    self._use(ws, pool_buf, addb, pn)
                    self._op(lib.pasn_se_gate_bwd, B(ws), B(pool_buf), Pm(se.fc1.weight), Pm(se.fc1.bias), Pm(se.fc2.weight), Pm(se.fc2.bias),
                             B(addb), B(pn), o[0], o[1], o[2], o[3], N, S, C, Cp, cse)
                    self._use(g.buf, y.buf, stat_buf, gate_buf, addb, ws, coef)
                    self._op(red, 2, B(g.buf), B(y.buf), stat, 0, B(gate_buf), B(addb), B(ws), B(coef), dg, db, N, S, C, Cp, actc, code, self.groups)
                if residual is not None and res_live:
                    self.add_grad(residual, g)  # after mode 0, g is the gradient of the pre-activation sum
                if not (x_live or w_live):
                    return  # only this unit's norm / bias / SE parameters were trainable: their gradients are out already
                if norm is not None and se_analytic:
                    self._use(g.buf, y.buf, stat_buf, coef, gate_buf, addb)
                    self._op(lib.pasn_bn_bwd_apply_se_g, B(g.buf), B(y.buf), stat, B(coef), B(gate_buf), B(addb), B(g.buf), N, S, C, Cp, code, self.groups)
                elif norm is not None:
                    dy = self.like(y) if residual is not None else g
                    self._use(g.buf, y.buf, stat_buf, coef, dy.buf)
                    self._op(lib.pasn_bn_bwd_apply_g, B(g.buf), B(y.buf), stat, B(coef), B(dy.buf), N, S, C, Cp, actc if lazy else 0, code, self.groups)
            elif not (x_live or w_live):
                return
            # ---- weight gradient
            dW = self.Gof(conv.weight)
            if kind == "first":
                if w_live:
                    wsz = int(lib.pasn_first_conv_wgrad_workspace_bytes(dref, code))
                    wsb = self._new_buf(wsz) if wsz else None
                    self._side_op(lib.pasn_first_conv_wgrad, (x.buf, dy.buf, wsb), B(x.buf), B(dy.buf), dW, dref, _lib.dtype_code(self.in_dtype), code, B(wsb))
                return
            if kind == "dw":
                if w_live:
                    wsb = self._new_buf(int(lib.pasn_dwconv3d_wgrad_workspace_floats(dref)) * 4)
                    self._side_op(lib.pasn_dwconv3d_wgrad, (x.buf, dy.buf, wsb), B(x.buf), B(dy.buf), B(wsb), dW, dref, code)
                if not x_live:
                    return
                dx = self.like(x)
                same = s == (1, 1, 1) and all(kk % 2 == 1 and pp == kk // 2 for kk, pp in zip(k, p))
                if same:
                    # stride-1 "same" depthwise conv: its input gradient is the forward stencil with the taps reversed
                    # (the T-marching forward kernel, not the generic gather)
                    wflip = torch.zeros(taps, Cp, dtype=torch.float32, device=self.device)
                    if self.native_pack and conv.weight.is_contiguous():
                        self.pack_jobs.append((conv.weight, wflip, 3, C, 1, taps, taps, Cp, 0, 0, 0))
                    else:
                        self.refresh.append(lambda: wflip[:, :C].copy_(conv.weight.detach().reshape(C, taps).flip(1).t()))
                    one_, zero_ = self.const(Cp, 1.0), self.const(Cp, 0.0)
                    dflip = self._desc(dy, dx, k, s, p, "none")
                    self.keep += [wflip]
                    hook = self.red_hook.get(x.buf)
                    rrows = int(lib.pasn_dwconv3d_dgrad_reduce_rows(ctypes.byref(dflip), code)) if hook is not None else 0
                    if rrows > 0 and self.readers.get(x.buf, 0) == 1 and self.grads.get(x.buf) is None:
                        # dx is the WHOLE gradient of the producer unit's output: its backward sums ride in this launch
                        wsr, hcoef = self._new_buf(x.N * rrows * 2 * x.Cp * 4), self._new_buf(2 * x.Cp * 4)
                        self._use(dy.buf, dx.buf, hook["y_buf"], hook["stat_buf"], wsr, hcoef)
                        self._op(lib.pasn_dwconv3d_dgrad_reduce, B(dy.buf), wflip.data_ptr(), one_.data_ptr(), zero_.data_ptr(), B(dx.buf),
                                 B(hook["y_buf"]), hook["stat"], hook["actc"], B(wsr), B(hcoef), hook["dg"](), hook["db"](),
                                 ctypes.byref(dflip), code)
                        hook["coef"] = hcoef
                    else:
                        self._use(dy.buf, dx.buf)
                        self._op(lib.pasn_dwconv3d_fwd, B(dy.buf), wflip.data_ptr(), one_.data_ptr(), zero_.data_ptr(), B(dx.buf), 0,
                                 ctypes.byref(dflip), code)
                else:
                    self._use(dy.buf, dx.buf)
                    self._op(lib.pasn_dwconv3d_dgrad, B(dy.buf), dw_w.data_ptr(), B(dx.buf), dref, code)
                self.add_grad(x, dx)
                return
            if w_live:
                wsz = int(lib.pasn_conv3d_wgrad_workspace_bytes(dref, code))  # windowed stride-1 convs, bf16: partial-buffer path
                wsb = self._new_buf(wsz) if wsz else None
                self._side_op(lib.pasn_conv3d_wgrad_ws, (x.buf, dy.buf, wsb), B(x.buf), B(dy.buf), dW, dref, code, B(wsb))
            if not x_live:
                return
            # ---- input gradient of the dense conv
            one = (1, 1, 1)
            have = self.grads.get(x.buf)
            if taps == 1 and p == (0, 0, 0):
                # 1x1x1: the same pointwise kernel with the transposed weight; strided ones on the compact grid, then scattered
                wt = lambda: conv.weight.detach().reshape(cout, cin).t()
                if s == one:
                    dx = self.like(x)
                    self._dense(dy, dx, one, one, (0, 0, 0), cin, cout, wt, residual=have, pack=(conv.weight, 1))
                    self.grads[x.buf] = dx
                else:
                    compact = Act(y.N, y.T, y.H, y.W, cin, x.Cp, self._new_buf(y.N * y.positions * x.Cp * self.es))
                    self._dense(dy, compact, one, one, (0, 0, 0), cin, cout, wt, pack=(conv.weight, 1))
                    dst = have if have is not None else self.like(x)
                    self._use(compact.buf, dst.buf)
                    self._op(lib.pasn_scatter_strided, B(compact.buf), B(dst.buf), dref, int(have is not None), code)
                    self.grads[x.buf] = dst
                return
            # windowed conv: correlate the (zero-inserted, for strides > 1) output gradient with the reversed, transposed weight,
            # padding k-1-p -- the forward implicit-GEMM kernel again.  Extent Z = Ti + 2p - k + 1 covers output_padding.
            wtf = lambda: (conv.weight.detach() if conv.weight.dim() == 5 else conv.weight.detach().unsqueeze(2)).transpose(0, 1).flip(2, 3, 4)
            padb = tuple(kk - 1 - pp for kk, pp in zip(k, p))
            Z = (x.T + 2 * p[0] - k[0] + 1, x.H + 2 * p[1] - k[1] + 1, x.W + 2 * p[2] - k[2] + 1)
            src = dy
            if s != one:
                zi = Act(y.N, Z[0], Z[1], Z[2], cout, y.Cp, self._new_buf(y.N * Z[0] * Z[1] * Z[2] * y.Cp * self.es))
                zd = ConvDesc(N=y.N, Ti=Z[0], Hi=Z[1], Wi=Z[2], Cin=cout, Cin_p=y.Cp, To=y.T, Ho=y.H, Wo=y.W, Cout=cout, Cout_p=y.Cp,
                              kt=1, kh=1, kw=1, st=s[0], sh=s[1], sw=s[2])
                self.keep.append(zd)
                self._use(dy.buf, zi.buf)
                self._op(lib.pasn_scatter_strided, B(dy.buf), B(zi.buf), ctypes.byref(zd), 0, code)
                src = zi
            else:
                assert (y.T, y.H, y.W) == Z
            dx = self.like(x)
            self._dense(src, dx, k, one, padb, cin, cout, wtf, residual=have, pack=(conv.weight, 1))
            self.grads[x.buf] = dx

        self.tape.append(backward)
        return out

    # ---- max pooling (ResNet-18 stem) ----------------------------------------------------------------------------------
    def maxpool_unit(self, x: Act, k, s, p) -> Act:
        y = self._out_act(x, x.C, k, s, p)
        d = self._desc(x, y, k, s, p, "none")
        dref, code, lib, B = ctypes.byref(d), self.code, self.lib, self.B
        self.readers[x.buf] = self.readers.get(x.buf, 0) + 1
        self._use(x.buf, y.buf)
        self._op(lib.pasn_maxpool3d_fwd, B(x.buf), B(y.buf), dref, code)
        self.live[y.buf] = self.live.get(x.buf, False)

        def backward() -> None:
            g = self.grads.get(y.buf)
            if g is None or not self.live[y.buf]:
                return
            dx = self.like(x)
            self._use(x.buf, g.buf, dx.buf)
            self._op(lib.pasn_maxpool3d_bwd, B(x.buf), B(g.buf), B(dx.buf), dref, code)
            self.add_grad(x, dx)

        self.tape.append(backward)
        return y

    # ---- head B tail ----------------------------------------------------------------------------------------------
    def xproto_tail(self, z: Optional[Act], r: Act, model, ext: Dict[str, int]) -> None:
        """z: add-on output (None = occurrence map only), r: occurrence-module output before the abs.  ``ext`` maps the names of
        the external tensors (occ, feat, sim, logits, dlogits, dsim, docc) to buffer ids filled in per run."""
        P, K = model.num_prototypes, model.num_classes
        D = model.prototype_shape[1]
        pv, fw = model.prototype_vectors, model.last_layer.weight
        for t in (pv, fw):
            _need_fp32_param(t, "a head parameter")
        d = XProtoDesc(N=r.N, S=r.positions, Cb=0, Cbp=0, D=D, Dp=(z.Cp if z is not None else round_up(D, 8)), Hd=D // 2,
                       Hp=round_up(D // 2, 8), P=P, Pp=r.Cp, K=K, mode=0 if z is not None else 1)
        self.keep.append(d)
        dref, code, lib, B, Pm = ctypes.byref(d), self.code, self.lib, self.B, self.Pm
        zb = z.buf if z is not None else None
        for rb_ in (zb, r.buf):
            if rb_ is not None:
                self.readers[rb_] = self.readers.get(rb_, 0) + 1
        e = ext
        wsz = int(lib.pasn_xproto_tail_workspace_bytes(dref))  # split-S pooling on the matrix cores (the inference head's kernels)
        wsb = self._new_buf(wsz) if wsz else None
        self._use(zb, r.buf, wsb)
        self._op(lib.pasn_xproto_tail_fwd_ws, B(zb), B(r.buf), Pm(pv), Pm(fw), B(e["occ"]), B(e["feat"]), B(e["sim"]), B(e["logits"]), dref, code,
                 B(wsb))

        def backward() -> None:
            dz = self.like(z) if z is not None else None
            dr = self.like(r)
            dfeat = self._new_buf(r.N * P * D * 4) if z is not None else None
            gp, gf = (self.Gof(pv, nullable=False), self.Gof(fw, nullable=False)) if z is not None else (0, 0)
            dzb = dz.buf if dz is not None else None
            self._use(zb, r.buf, dzb, dr.buf, dfeat)
            self._op(lib.pasn_xproto_tail_bwd, B(zb), B(r.buf), Pm(pv), Pm(fw), B(e["feat"]), B(e["sim"]), B(e["dlogits"]), B(e["dsim"]),
                     B(e["docc"]), B(dfeat), B(dzb), B(dr.buf), gp, gf, dref, code)
            if z is not None and self.live.get(z.buf, False):
                self.add_grad(z, dz)
            if self.live.get(r.buf, False):
                self.add_grad(r, dr)

        self.tape.append(backward)

    # ---- head A tail (ProtoPNet) -----------------------------------------------------------------------------------
    def l2_tail(self, z: Act, model, ext: Dict[str, int]) -> None:
        """z: add-on output after the Sigmoid.  External tensors: min_d, logits (forward), dlogits, dmin (backward)."""
        P, K, D = model.num_prototypes, model.num_classes, model.prototype_shape[1]
        pv, fw = model.prototype_vectors, model.last_layer.weight
        for t in (pv, fw):
            _need_fp32_param(t, "a head parameter")
        act = model.prototype_activation_function
        if act not in ("log", "linear"):
            raise NotImplementedError("only the 'log' and 'linear' prototype activations run on the HIP path")
        actc, eps = (0 if act == "log" else 1), float(model.epsilon)
        N, S = z.N, z.positions
        code, lib, B, Pm, e = self.code, self.lib, self.B, self.Pm, ext
        amin = self._new_buf(N * P * 4)
        self.readers[z.buf] = self.readers.get(z.buf, 0) + 1
        self._use(z.buf, amin)
        self._op(lib.pasn_l2_head_fwd, B(z.buf), Pm(pv), Pm(fw), 0, B(e["min_d"]), B(amin), B(e["logits"]), N, S, D, z.Cp, P, K, code, actc, eps)

        def backward() -> None:
            dz = self.like(z)
            coef = self._new_buf(N * P * 4)
            self._use(z.buf, amin, dz.buf, coef)
            self._op(lib.pasn_l2_head_bwd, B(z.buf), Pm(pv), Pm(fw), B(e["min_d"]), B(amin), B(e["dlogits"]), B(e["dmin"]), B(dz.buf), B(coef),
                     self.Gof(pv, nullable=False), self.Gof(fw, nullable=False), N, S, D, z.Cp, P, K, code, actc, eps)
            if self.live.get(z.buf, False):
                self.add_grad(z, dz)

        self.tape.append(backward)

    # ---- finish: generate the backward list, then place every buffer ----------------------------------------------------
    def finish_train(self, x_in: Act, ext: Dict[str, int]) -> "TrainPlan":
        self.n_fwd = len(self.ops)
        for emit in reversed(self.tape):
            emit()
        while self._forks:
            self._join()
        live: List[Tuple[int, int, int]] = []
        total = 0
        for b in self.bufs:
            if b.external:
                continue
            live = [a for a in live if a[2] >= b.first]
            live.sort()
            off = 0
            for (o, sz, _) in live:
                if off + b.nbytes <= o:
                    break
                off = max(off, o + sz)
            b.offset = off
            live.append((off, b.nbytes, b.last))
            total = max(total, off + b.nbytes)
        return TrainPlan(self, x_in, ext, total)


def build_pack_tables(jobs: List[tuple], device) -> tuple:
    """Device tables of ``pasn_pack_weights`` for jobs (parameter, destination, mode, cout, cin, taps, rows, kc, frag, kstep, ch):
    (job structs as bytes, job of each block, chunk of each block, number of blocks)."""
    import numpy as np

    chunk = int(_lib.lib().pasn_pack_chunk())
    jt = np.zeros(len(jobs), dtype=np.dtype(
        [("src", "u8"), ("dst", "u8"), ("n", "i8"), ("mode", "i4"), ("cout", "i4"), ("cin", "i4"), ("taps", "i4"), ("rows", "i4"),
         ("kc", "i4"), ("frag", "i4"), ("bf16", "i4"), ("kstep", "i4"), ("ch", "i4")], align=True))
    assert jt.dtype.itemsize == 64  # struct pasn_pack_job
    bj, bc = [], []
    for i, (param, dst, mode, cout, cin, taps, rows, kc, frag, kstep, ch) in enumerate(jobs):
        assert param.dtype == torch.float32 and param.is_contiguous() and dst.is_contiguous()
        jt[i] = (param.data_ptr(), dst.data_ptr(), dst.numel(), mode, cout, cin, taps, rows, kc, frag, int(dst.dtype == torch.bfloat16),
                 kstep, ch)
        nb = (dst.numel() + chunk - 1) // chunk
        bj += [i] * nb
        bc += list(range(nb))
    return (torch.from_numpy(jt.view(np.uint8).copy()).to(device), torch.tensor(bj, dtype=torch.int32, device=device),
            torch.tensor(bc, dtype=torch.int32, device=device), len(bj))


class TrainPlan:
    def __init__(self, tb: TrainBuilder, x_in: Act, ext: Dict[str, int], arena_bytes: int):
        self.ops, self.n_fwd, self.keep, self.refresh, self.op_names = tb.ops, tb.n_fwd, tb.keep, tb.refresh, tb.op_names
        self.op_bytes, self.op_kind, self.op_join = tb.op_bytes, tb.op_kind, tb.op_join
        self.serial = not any(k == 1 for k in tb.op_kind)  # True: every launch on the caller's stream (also set by the per-launch profilers)
        self._side = None
        self._events: Dict[int, "torch.cuda.Event"] = {}
        self.offsets = [None if b.external else b.offset for b in tb.bufs]
        self.in_buf, self.ext, self.gbuf = x_in.buf, ext, tb.gbuf
        self.pslots, self.gsize, self.nbt, self.groups = tb.pslots, tb.gsize, tb.nbt, tb.groups
        self.arena_bytes = arena_bytes
        self.naive_bytes = sum(b.nbytes for b in tb.bufs if not b.external)
        self.device, self.dtype = tb.device, tb.dtype
        self.pack_jobs, self._pack_key, self._pack_tables = tb.pack_jobs, None, None

    def _pack_weights(self, st) -> None:
        """All conv weights of the step from the live parameters, one launch (csrc/pack.hip).  The job table holds device pointers: it is
        rebuilt when a parameter's storage has moved (``.to()``, ``.data = ...``); in-place optimizer updates keep it."""
        if not self.pack_jobs:
            return
        key = tuple(j[0].data_ptr() for j in self.pack_jobs)
        if key != self._pack_key:
            self._pack_tables = build_pack_tables(self.pack_jobs, self.device)
            self._pack_key = key
        jobs, bj, bc, nb = self._pack_tables
        _lib.check(_lib.lib().pasn_pack_weights(jobs.data_ptr(), bj.data_ptr(), bc.data_ptr(), nb, st))

    def _ptrs(self, arena: torch.Tensor) -> List[int]:
        base = round_up(arena.data_ptr(), ALIGN)
        return [0 if o is None else base + o for o in self.offsets]

    def forward(self, x: torch.Tensor, outs: Dict[str, torch.Tensor]):
        with torch.no_grad():
            for r in self.refresh:
                r()
            if self.nbt:
                torch._foreach_add_(self.nbt, self.groups)
        self._pack_weights(_lib.current_stream())
        arena = torch.empty(self.arena_bytes + ALIGN, dtype=torch.uint8, device=x.device)
        ptrs = self._ptrs(arena)
        ptrs[self.in_buf] = x.data_ptr()
        for name, t in outs.items():
            ptrs[self.ext[name]] = 0 if t is None else t.data_ptr()
        st = _lib.current_stream()
        for op in self.ops[: self.n_fwd]:
            op(ptrs, st)
        return arena

    def backward(self, arena: torch.Tensor, x: torch.Tensor, tensors: Dict[str, Optional[torch.Tensor]]) -> List[torch.Tensor]:
        """``tensors``: the forward outputs plus dlogits / dsim / docc (None = no gradient).  Returns one gradient per
        ``self.pslots`` entry (views of one flat fp32 buffer -- the bucket a data-parallel all-reduce can send as is)."""
        G = torch.zeros(max(self.gsize, 1), dtype=torch.float32, device=x.device)
        ptrs = self._ptrs(arena)
        ptrs[self.in_buf] = x.data_ptr()
        ptrs[self.gbuf] = G.data_ptr()
        for name, t in tensors.items():
            ptrs[self.ext[name]] = 0 if t is None else t.data_ptr()
        st = _lib.current_stream()
        if self.serial:
            for op in self.ops[self.n_fwd:]:
                op(ptrs, st)
        else:
            main = torch.cuda.current_stream(x.device)
            if self._side is None:
                self._side = torch.cuda.Stream(x.device)
            side, sst, ev = self._side, self._side.cuda_stream, self._events
            for i in range(self.n_fwd, len(self.ops)):
                kind = self.op_kind[i]
                if kind == 0:
                    self.ops[i](ptrs, st)
                elif kind == 1:
                    # the side launch reads what the main stream has produced so far (its dy); the join below is where the main stream
                    # may first overwrite what it reads
                    before = ev.get(-i - 1) or ev.setdefault(-i - 1, torch.cuda.Event())
                    before.record(main)
                    side.wait_event(before)
                    self.ops[i](ptrs, sst)
                    done = ev.get(i) or ev.setdefault(i, torch.cuda.Event())
                    done.record(side)
                else:
                    main.wait_event(ev[self.op_join[i]])
        return [G[o: o + n].view_as(p) for (p, o, n) in self.pslots]


class _TrainFn(torch.autograd.Function):
    """Autograd node of one training-mode pass: forward replays the forward launch list, backward the backward list."""

    @staticmethod
    def forward(ctx, runner, x, *params):
        outs = runner.alloc_outputs(x)
        arena = runner.plan.forward(x, outs)
        ctx.runner, ctx.names, ctx.consumed = runner, tuple(outs), False
        ctx.save_for_backward(x, arena, *outs.values())
        ctx.set_materialize_grads(False)
        return tuple(outs[n] for n in runner.out_names)

    @staticmethod
    def backward(ctx, *grads):
        runner = ctx.runner
        if ctx.consumed:
            # the backward launch list runs IN the forward's arena: activations are overwritten as soon as their last reader has
            # run and gradient buffers are rewritten in place, so a second replay would read clobbered data and return wrong
            # gradients silently (retain_graph=True, or autograd.grad followed by backward)
            raise RuntimeError("protoasnet_amd: this training pass was already differentiated once; its activations were reused as "
                               "backward scratch.  retain_graph / a second backward through the same forward is not supported -- "
                               "run the forward again")
        ctx.consumed = True
        x, arena, *saved = ctx.saved_tensors
        outs = dict(zip(ctx.names, saved))
        tensors: Dict[str, Optional[torch.Tensor]] = dict(outs)
        for name, g in zip(runner.grad_names, grads):
            tensors[name] = None if g is None else g.contiguous().float()
        tensors = runner.fix_grads(tensors, outs)
        pg = runner.plan.backward(arena, x, tensors)
        by_id = {id(p): g for (p, _, _), g in zip(runner.plan.pslots, pg)}
        return (None, None) + tuple(by_id.get(id(p)) for p in runner.params)


class TrainRunner:
    """One compiled training pass of a model for one input shape.  Head B (XProtoNet / Video_XProtoNet): mode 0 = forward(),
    1 = compute_occurence_map().  Head A (PPNet): mode 0 = forward() -> (logits, min_distances)."""

    def __init__(self, model, x: torch.Tensor, mode: int, head: str = "B"):
        dtype = model._dtype()
        # mode 2 (head B): mode 0 over [clips, warped clips] -- two statistics groups, the first half's (logits, similarity, occurrence map)
        # and the second half's occurrence map are what the reference's two passes (forward + compute_occurence_map) return
        tb = TrainBuilder(x.device, dtype, x.dtype, groups=2 if mode == 2 else 1)
        x_in = tb.input(tuple(x.shape))
        trunk = model.cnn_backbone if head == "B" else model.features
        feat = trunk.build_train(tb, x_in)
        self.head, self.mode, self.model = head, mode, model
        self.N, self.S = feat.N, feat.positions
        self.spatial = (feat.T, feat.H, feat.W) if x.dim() == 5 else (feat.H, feat.W)
        if head == "B":
            ext = {n: tb._new_buf(0, external=True) for n in ("occ", "feat", "sim", "logits", "dlogits", "dsim", "docc")}
            z = None
            if mode in (0, 2):
                z = feat
                for conv, act in model.add_on_layers._steps():
                    z = tb.unit(z, conv, None, act)
            r = feat
            for conv, act in model.occurrence_module._steps():
                r = tb.unit(r, conv, None, act)
            tb.xproto_tail(z, r, model, ext)
            self.out_names = ("logits", "sim", "occ") if mode in (0, 2) else ("occ",)
            self.grad_names = ("dlogits", "dsim", "docc") if mode in (0, 2) else ("docc",)
        else:
            ext = {n: tb._new_buf(0, external=True) for n in ("min_d", "logits", "dlogits", "dmin")}
            z = feat
            for conv, act in model.add_on_layers._steps():
                z = tb.unit(z, conv, None, act)
            tb.l2_tail(z, model, ext)
            self.out_names, self.grad_names = ("logits", "min_d"), ("dlogits", "dmin")
        self.plan = tb.finish_train(x_in, ext)
        self.params = [p for p in model.parameters()]

    def alloc_outputs(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        m, dev, f32 = self.model, x.device, torch.float32
        P, D, K = m.num_prototypes, m.prototype_shape[1], m.num_classes
        if self.head == "A":
            return {"logits": torch.empty((self.N, K), dtype=f32, device=dev), "min_d": torch.empty((self.N, P), dtype=f32, device=dev)}
        outs = {"occ": torch.empty((self.N, P, 1) + self.spatial, dtype=f32, device=dev)}
        if self.mode in (0, 2):
            outs["feat"] = torch.empty((self.N, P, D), dtype=f32, device=dev)
            outs["sim"] = torch.empty((self.N, P), dtype=f32, device=dev)
            outs["logits"] = torch.empty((self.N, K), dtype=f32, device=dev)
        else:
            outs["feat"] = outs["sim"] = outs["logits"] = None
        return outs

    def fix_grads(self, tensors, outs):
        if self.head == "A":
            if tensors.get("dlogits") is None:
                tensors["dlogits"] = torch.zeros_like(outs["logits"])
            tensors.setdefault("dmin", None)
            return tensors
        if self.mode in (0, 2) and tensors.get("dlogits") is None:
            tensors["dlogits"] = torch.zeros_like(outs["logits"])
        if self.mode == 1:
            tensors.setdefault("dlogits", None)
            tensors.setdefault("dsim", None)
            if tensors.get("docc") is None:
                tensors["docc"] = torch.zeros_like(outs["occ"])
        return tensors

    def __call__(self, x: torch.Tensor):
        return _TrainFn.apply(self, x, *self.params)
