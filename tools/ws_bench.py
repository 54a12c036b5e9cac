#!/usr/bin/env python3
"""A/B of the weight-stationary pointwise kernel (pwconv_ws.hip) against the kernels it replaces, per X3D-S / head-B layer shape.

    python tools/ws_bench.py [reps]        # N = 32 clips, bf16; prints one line per layer: old kernel us, new kernel us, GB/s, max |diff|

Interleaved rounds in one process (guide rule 24); env knobs of the new kernel (PASN_WS_PT / _MT / _NS / _BPC) apply to its arm."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protoasnet_amd import _lib  # noqa: E402
import torch.nn as nn

from protoasnet_amd.plan import Act, PlanBuilder, round_up

DEV = torch.device("cuda")
LAYERS = [
    # cin, cout, T, H, W, residual, gate
    (54, 24, 16, 56, 56, True, True), (54, 24, 16, 56, 56, True, False),
    (48, 108, 16, 28, 28, False, False), (108, 48, 16, 28, 28, True, True), (108, 48, 16, 28, 28, True, False),
    (48, 216, 16, 28, 28, False, False),
    (96, 216, 16, 14, 14, False, False), (216, 96, 16, 14, 14, True, True), (216, 96, 16, 14, 14, True, False),
    (96, 432, 16, 14, 14, False, False),
    (192, 432, 16, 7, 7, False, False), (432, 192, 16, 7, 7, True, True), (432, 192, 16, 7, 7, True, False),
    (192, 256, 16, 7, 7, False, False), (256, 256, 16, 7, 7, False, False), (256, 128, 16, 7, 7, False, False), (128, 30, 16, 7, 7, False, False),
]


def build(cin, cout, T, H, W, use_res, gate, ws, N=32, dtype=torch.bfloat16):
    os.environ["PASN_WS"] = "1" if ws else "0"
    _lib.tuning_reload()  # the library routes on one snapshot of the switches
    torch.manual_seed(1)
    pb = PlanBuilder(DEV, dtype, dtype)
    cp, cop = round_up(cin, 8), round_up(cout, 8)
    x = torch.zeros(N, T, H, W, cp, device=DEV, dtype=dtype)
    x[..., :cin] = torch.randn(N, T, H, W, cin, device=DEV).to(dtype)
    xa = Act(N, T, H, W, cin, cp, pb._new_buf(x.numel() * 2, external=True))
    conv = nn.Conv3d(cin, cout, 1, bias=False).to(DEV)
    bn = nn.BatchNorm3d(cout).to(DEV).eval()
    ra = rt = gb = gt = None
    if use_res:
        rt = torch.zeros(N, T, H, W, cop, device=DEV, dtype=dtype)
        rt[..., :cout] = torch.randn(N, T, H, W, cout, device=DEV).to(dtype)
        ra = Act(N, T, H, W, cout, cop, pb._new_buf(rt.numel() * 2, external=True))
    if gate:
        gt = torch.zeros(N, cp, device=DEV)
        gt[:, :cin] = torch.rand(N, cin, device=DEV)
        gb = pb._new_buf(gt.numel() * 4, external=True)
    y = pb.conv(xa, conv, bn, "relu", residual=ra, in_gate=gb, in_swish=gate)
    plan = pb.finish(xa, y)
    if ra is not None:
        plan.ptrs[ra.buf] = rt.data_ptr()
    if gb is not None:
        plan.ptrs[gb] = gt.data_ptr()
    return plan, x, (rt, gt, conv, bn)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = os.environ.get("WS_ONLY")
    for L in LAYERS:
        if only and f"{L[0]}->{L[1]}" not in only.split(","):
            continue
        arms = {}
        try:
            for ws in (False, True):
                arms[ws] = build(*L, ws)
        except Exception as e:  # a layer one arm cannot take
            print(f"{L}: {e}", flush=True)
            continue
        outs, times = {}, {False: [], True: []}
        for ws in (False, True):
            plan, x, _ = arms[ws]
            os.environ["PASN_WS"] = "1" if ws else "0"
            _lib.tuning_reload()  # the library routes on one snapshot of the switches
            for _ in range(3):
                outs[ws] = plan.run(x)
        torch.cuda.synchronize()
        for rnd in range(5):
            for ws in (False, True):
                plan, x, _ = arms[ws]
                os.environ["PASN_WS"] = "1" if ws else "0"
                _lib.tuning_reload()  # the library routes on one snapshot of the switches
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    plan.run(x)
                e1.record()
                torch.cuda.synchronize()
                times[ws].append(e0.elapsed_time(e1) / reps * 1e3)
        diff = float((outs[True].float() - outs[False].float()).abs().max())
        mo, mn = arms[False][0].meta[0], arms[True][0].meta[0]
        to, tn = sorted(times[False])[2], sorted(times[True])[2]
        print(f"{L[0]:4d}->{L[1]:<4d} {L[2]}x{L[3]}x{L[4]} res={int(L[5])} gate={int(L[6])}  {mo['kernel']:42s} {to:7.1f} us | {mn['kernel']:34s} {tn:7.1f} us (min {min(times[True]):6.1f})"
              f"  {mn['bytes'] / 1e6:7.1f} MB {mn['bytes'] / tn / 1e3:6.0f} GB/s  x{to / tn:4.2f}  maxdiff {diff:.3g}", flush=True)


if __name__ == "__main__":
    main()
