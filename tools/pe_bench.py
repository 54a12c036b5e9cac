#!/usr/bin/env python3
"""The streamed project + expand pair (x3d_pe.hip) at the X3D-S benchmark shape (32 x 16 x 7 x 7, 432 -> 192 -> 432) against the launches it replaces.

    python tools/pe_bench.py [reps]
    PASN_LIB_PATH=.../libprotoasnet_amd_tuning.so PASN_PE_STAMPS=1 python tools/pe_bench.py     # + in-kernel phase stamps of block 0
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


def build(se, streamed, N=32, T=16, hw=7, cm=432, c=192):
    with _lib.tuning_env(PASN_NO_PE=None if streamed else "1"):
        torch.manual_seed(3)
        pb = PlanBuilder(DEV, torch.bfloat16, torch.bfloat16)
        x = torch.relu(torch.randn(N, T, hw, hw, cm, device=DEV)).bfloat16()
        r = torch.relu(torch.randn(N, T, hw, hw, c, device=DEV)).bfloat16()
        xa = Act(N, T, hw, hw, cm, cm, pb._new_buf(x.numel() * 2, external=True))
        ra = Act(N, T, hw, hw, c, c, pb._new_buf(r.numel() * 2, external=True))
        m = [nn.Conv3d(cm, cm, 3, 1, 1, groups=cm, bias=False), nn.BatchNorm3d(cm), nn.Conv3d(cm, c, 1, bias=False), nn.BatchNorm3d(c),
             nn.Conv3d(c, cm, 1, bias=False), nn.BatchNorm3d(cm), nn.Conv3d(cm, 32, 1), nn.Conv3d(32, cm, 1)]
        m = [mm.to(DEV).eval() for mm in m]
        if se:
            y, pooled = pb.dwconv(xa, m[0], m[1], act="none", pool=True)
            pair = pb.conv_pair(y, m[2], m[3], "relu", ra, m[4], m[5], "relu", in_swish=True, se=(pooled, m[6], m[7]))
            if pair is None:
                o1 = pb.conv_se(y, m[2], m[3], "relu", ra, pooled, m[6], m[7])
                o2 = pb.conv(o1, m[4], m[5], "relu")
            else:
                o1, o2 = pair
        else:
            y = pb.dwconv(xa, m[0], m[1], act="swish")
            pair = pb.conv_pair(y, m[2], m[3], "relu", ra, m[4], m[5], "relu")
            if pair is None:
                o1 = pb.conv(y, m[2], m[3], "relu", residual=ra)
                o2 = pb.conv(o1, m[4], m[5], "relu")
            else:
                o1, o2 = pair
        plan = pb.finish(xa, o2)
        plan.ptrs[ra.buf] = r.data_ptr()
        names = [mm["kernel"] for mm in pb.meta]
    return plan, x, names


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for se in (False, True):
        arms = {f: build(se, f) for f in (True, False)}
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
        print(f"se={se}: stencil + streamed pair {min(times[True]):6.1f} us ({' + '.join(arms[True][2])})   separate {min(times[False]):6.1f} us ({' + '.join(arms[False][2])})", flush=True)
        if os.environ.get("PASN_PE_STAMPS"):
            h = ctypes.CDLL(_lib.LIB_PATH)
            buf = (ctypes.c_longlong * 8)()
            arms[True][0].run(arms[True][1])
            torch.cuda.synchronize()
            h.pasn_debug_pe_stamps(buf)
            v = [b / 100.0 for b in buf]
            print(f"  block 0 (us): tables {v[1] - v[0]:.2f}  rows landed + gate {v[2] - v[1]:.2f}  transform {v[3] - v[2]:.2f}  project {v[4] - v[3]:.2f}  expand {v[5] - v[4]:.2f}  total {v[5] - v[0]:.2f}")


if __name__ == "__main__":
    main()
