#!/usr/bin/env python3
"""The fused residual-block launch (x3d_block.hip) against the launches it replaces, at the X3D-S benchmark shapes (N = 32, T = 16, bf16).

    python tools/block_bench.py [reps]                       # per stage: fused us, separate us (stencil + project / pair + expand), max |diff|
    PASN_LIB_PATH=.../libprotoasnet_amd_tuning.so PASN_BLOCK_ABL=<bits> python tools/block_bench.py    # timing ablations of the fused launch
        bits: 1 stencil MFMAs, 2 frame DMA, 4 project phase, 8 expand phase (results are wrong when set)

Interleaved rounds in one process; every arm is a compiled plan (so the numbers include nothing but the launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from protoasnet_amd import _lib
from protoasnet_amd.plan import Act, PlanBuilder, round_up

DEV = torch.device("cuda")
STAGES = [("stage3", 108, 48, 108, 28), ("stage4", 216, 96, 216, 14), ("stage5", 432, 192, 432, 7)]


def mods(cm, c, cn):
    torch.manual_seed(cm)
    conv_b = nn.Conv3d(cm, cm, 3, 1, 1, groups=cm, bias=False)
    conv_c = nn.Conv3d(cm, c, 1, bias=False)
    conv_a = nn.Conv3d(c, cn, 1, bias=False)
    bns = [nn.BatchNorm3d(cm), nn.BatchNorm3d(c), nn.BatchNorm3d(cn)]
    for bn in bns:
        bn.eval()
        with torch.no_grad():
            bn.running_var.uniform_(0.5, 1.5)
            bn.running_mean.normal_(0, 0.2)
    return [m.to(DEV) for m in (conv_b, bns[0], conv_c, bns[1], conv_a, bns[2])]


def build(cm, c, cn, hw, fused, N=32, T=16):
    with _lib.tuning_env(PASN_BLOCK="1" if fused else "0"):
        pb = PlanBuilder(DEV, torch.bfloat16, torch.bfloat16)
        cmp_, cp = round_up(cm, 8), round_up(c, 8)
        e = torch.zeros(N, T, hw, hw, cmp_, device=DEV, dtype=torch.bfloat16)
        e[..., :cm] = torch.relu(torch.randn(N, T, hw, hw, cm, device=DEV)).bfloat16()
        r = torch.relu(torch.randn(N, T, hw, hw, cp, device=DEV)).bfloat16()
        ea = Act(N, T, hw, hw, cm, cmp_, pb._new_buf(e.numel() * 2, external=True))
        ra = Act(N, T, hw, hw, c, cp, pb._new_buf(r.numel() * 2, external=True))
        m = mods(cm, c, cn)
        if fused:
            y, en = pb.x3d_block(ea, m[0], m[1], m[2], m[3], ra, m[4], m[5])
        else:
            d = pb.dwconv(ea, m[0], m[1], act="swish")
            pair = pb.conv_pair(d, m[2], m[3], "relu", ra, m[4], m[5], "relu")
            if pair is not None:
                y, en = pair
            else:
                y = pb.conv(d, m[2], m[3], act="relu", residual=ra)
                en = pb.conv(y, m[4], m[5], act="relu")
        pb.bufs[y.buf].external = True
        plan = pb.finish(ea, en)
        yout = torch.empty(N, T, hw, hw, cp, device=DEV, dtype=torch.bfloat16)
        plan.ptrs[y.buf] = yout.data_ptr()
        plan.ptrs[ra.buf] = r.data_ptr()
        names = [mm["kernel"] for mm in pb.meta]
    return plan, e, yout, names


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = os.environ.get("BLOCK_STAGE")
    for name, cm, c, cn, hw in STAGES:
        if only and only != name:
            continue
        arms = {f: build(cm, c, cn, hw, f) for f in (True, False)}
        outs = {}
        for f in (True, False):
            plan, e, yout, _ = arms[f]
            for _ in range(3):
                o = plan.run(e)
            torch.cuda.synchronize()
            outs[f] = (o.clone(), yout.clone())
        times = {True: [], False: []}
        for _ in range(5):
            for f in (True, False):
                plan, e, _, _ = arms[f]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    plan.run(e)
                e1.record()
                torch.cuda.synchronize()
                times[f].append(e0.elapsed_time(e1) * 1e3 / reps)
        if int(os.environ.get("PASN_BLOCK_ABL", "0")) & 64:
            import ctypes

            h = ctypes.CDLL(_lib.LIB_PATH)
            buf = (ctypes.c_longlong * 136)()
            arms[True][0].run(arms[True][1])
            torch.cuda.synchronize()
            h.pasn_debug_block_stamps(buf, 136)
            v = list(buf)
            us = lambda c: c / 100.0  # s_memrealtime ticks at 100 MHz
            print(f"  stamps (block 0, us): clear {us(v[1] - v[0]):.2f}")
            for t in range(8):
                b = 8 + 8 * t
                if v[b] == 0:
                    break
                pk = v[b + 6]
                print(f"  tile {t}: D {us(v[b + 1] - v[b]):.2f} (waits+barriers {us(v[b + 4]):.2f}, issue+build {us(pk & 0xfffff):.2f}, mfma {us((pk >> 20) & 0xfffff):.2f}, "
                      f"emit {us((pk >> 40) & 0xfffff):.2f}; {v[b + 5]} steps)  P {us(v[b + 2] - v[b + 1]):.2f}  E {us(v[b + 3] - v[b + 2]):.2f}"
                      f"  (tile start {us(v[b] - v[0]):.2f})")
        diff = max(float((outs[True][0].float() - outs[False][0].float()).abs().max()), float((outs[True][1].float() - outs[False][1].float()).abs().max()))
        print(f"{name}: fused {min(times[True]):7.1f} us   separate {min(times[False]):7.1f} us ({' + '.join(arms[False][3])})   max|diff| {diff:.3g}"
              f"   tuning: {_lib.tuning_report().strip() or '-'}", flush=True)


if __name__ == "__main__":
    main()
