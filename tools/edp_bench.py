#!/usr/bin/env python3
"""The whole-block launch of the 7 x 7 stage (x3d_edp.hip) at the X3D-S benchmark shape (32 x 16 x 7 x 7, 192 -> 432 -> 192 -> 432) against the
launches it replaces.

    python tools/edp_bench.py [reps]
    PASN_LIB_PATH=.../libprotoasnet_amd_tuning.so PASN_EDP_STAMPS=1 python tools/edp_bench.py     # + in-kernel phase stamps of block 0
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from protoasnet_amd import _lib
from protoasnet_amd.plan import Act, PlanBuilder

DEV = torch.device("cuda")


def build(whole, N=32, T=16, cx=192, cm=432):
    with _lib.tuning_env(PASN_NO_EDP=None if whole else "1"):
        torch.manual_seed(3)
        pb = PlanBuilder(DEV, torch.bfloat16, torch.bfloat16)
        x = torch.relu(torch.randn(N, T, 7, 7, cx, device=DEV)).bfloat16()
        xa = Act(N, T, 7, 7, cx, cx, pb._new_buf(x.numel() * 2, external=True))
        m = [nn.Conv3d(cx, cm, 1, bias=False), nn.BatchNorm3d(cm), nn.Conv3d(cm, cm, 3, 1, 1, groups=cm, bias=False), nn.BatchNorm3d(cm),
             nn.Conv3d(cm, cx, 1, bias=False), nn.BatchNorm3d(cx), nn.Conv3d(cx, cm, 1, bias=False), nn.BatchNorm3d(cm)]
        m = [mm.to(DEV).eval() for mm in m]
        if whole:
            y, en = pb.x3d_edp(xa, *m)
        else:
            e = pb.conv(xa, m[0], m[1], act="relu")
            d = pb.dwconv(e, m[2], m[3], act="swish")
            y, en = pb.conv_pair(d, m[4], m[5], "relu", xa, m[6], m[7], "relu")
        plan = pb.finish(xa, en)
        names = [mm["kernel"] for mm in pb.meta]
    return plan, x, names


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    arms = {f: build(f) for f in (True, False)}
    times = {True: [], False: []}
    for f in (True, False):
        for _ in range(3):
            arms[f][0].run(arms[f][1])
    torch.cuda.synchronize()
    for _ in range(5):
        for f in (True, False):
            plan, x, _ = arms[f]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                plan.run(x)
            e1.record()
            torch.cuda.synchronize()
            times[f].append(e0.elapsed_time(e1) * 1e3 / reps)
    print(f"whole block {min(times[True]):6.1f} us ({' + '.join(arms[True][2])})   separate {min(times[False]):6.1f} us ({' + '.join(arms[False][2])})", flush=True)
    if os.environ.get("PASN_EDP_STAMPS"):
        h = ctypes.CDLL(_lib.LIB_PATH)
        buf = (ctypes.c_longlong * 8)()
        arms[True][0].run(arms[True][1])
        torch.cuda.synchronize()
        h.pasn_debug_edp_stamps(buf)
        v = [b / 100.0 for b in buf]
        print(f"  block 0 (us): clear + x image {v[1] - v[0]:.2f}  quads {v[2] - v[1]:.2f} (expand {v[5]:.2f}, stencil {v[6]:.2f}, project {v[7]:.2f})  "
              f"project epilogue {v[3] - v[2]:.2f}  next expand {v[4] - v[3]:.2f}  total {v[4] - v[0]:.2f}")


if __name__ == "__main__":
    main()
