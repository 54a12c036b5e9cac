#!/usr/bin/env python3
"""The fused expand + stencil launch at stride 1 (x3d_expdw_tz.hip, Toeplitz form, against x3d_expdw.hip, block-diagonal form, and the two
separate launches) at the X3D-S benchmark shapes (N = 32, T = 16, bf16).

    python tools/tz_bench.py [reps]
    PASN_LIB_PATH=.../libprotoasnet_amd_tuning.so PASN_TZ_STAMPS=1 python tools/tz_bench.py     # + in-kernel phase stamps of block 0 / wave 0
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


def build(arm, se, cin, cm, hw, N=32, T=16):
    env = {"tz": dict(PASN_EXPDW_TZ=None), "bd": dict(PASN_EXPDW_TZ="0"), "two": dict(PASN_EXPDW="0")}[arm]
    with _lib.tuning_env(**env):
        torch.manual_seed(3)
        pb = PlanBuilder(DEV, torch.bfloat16, torch.bfloat16)
        x = torch.relu(torch.randn(N, T, hw, hw, cin, device=DEV)).bfloat16()
        xa = Act(N, T, hw, hw, cin, cin, pb._new_buf(x.numel() * 2, external=True))
        m = [nn.Conv3d(cin, cm, 1, bias=False), nn.BatchNorm3d(cm), nn.Conv3d(cm, cm, 3, 1, 1, groups=cm, bias=False), nn.BatchNorm3d(cm)]
        m = [mm.to(DEV).eval() for mm in m]
        act = "none" if se else "swish"
        out = pb.expand_dw(xa, m[0], m[1], m[2], m[3], act, pool=se) if arm != "two" else None
        if out is None:
            e = pb.conv(xa, m[0], m[1], act="relu")
            out = pb.dwconv(e, m[2], m[3], act=act, pool=se)
        y = out[0] if se else out
        pool_t = None
        if se:
            pb.bufs[out[1][0]].external = True
        plan = pb.finish(xa, y)
        if se:
            pool_t = torch.zeros(N, out[1][1], y.Cp, dtype=torch.float32, device=DEV)
            plan.ptrs[out[1][0]] = pool_t.data_ptr()
        names = [mm["kernel"] for mm in pb.meta]
    return plan, x, names, pool_t


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    shapes = [(24, 54, 56)] + ([(48, 108, 28)] if os.environ.get("TZ_BENCH_ALL") else [])
    for cin, cm, hw in shapes:
        for se in (False, True):
            arms = {a: build(a, se, cin, cm, hw) for a in ("tz", "bd", "two")}
            times = {a: [] for a in arms}
            for a in arms:
                for _ in range(3):
                    arms[a][0].run(arms[a][1])
            torch.cuda.synchronize()
            for _ in range(5):
                for a in arms:
                    plan, x = arms[a][0], arms[a][1]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        plan.run(x)
                    e1.record()
                    torch.cuda.synchronize()
                    times[a].append(e0.elapsed_time(e1) * 1e3 / reps)
            print(f"{cin}->{cm} {hw}x{hw} se={se}: " + "   ".join(f"{a} {min(times[a]):6.1f} us ({' + '.join(arms[a][2])})" for a in arms), flush=True)
            if os.environ.get("PASN_TZ_STAMPS") and arms["tz"][2][0].startswith("x3d_expdw_tz"):
                h = ctypes.CDLL(_lib.LIB_PATH)
                buf = (ctypes.c_longlong * 62)()
                arms["tz"][0].run(arms["tz"][1])
                torch.cuda.synchronize()
                h.pasn_debug_tz_stamps(buf)
                v = list(buf)
                print(f"  block {int(os.environ['PASN_TZ_STAMPS']) - 1} / wave 0 (shader cycles): operands {v[1] - v[0]}  prologue {v[2] - v[1]}")
                for k in range(8):
                    b = 3 + 6 * k
                    prev = v[2] if k == 0 else v[b - 1]
                    print(f"  step {k}: stencil {v[b] - prev:6d}  barrier {v[b + 1] - v[b]:6d}  store {v[b + 2] - v[b + 1]:6d}  expand {v[b + 3] - v[b + 2]:6d}  loads {v[b + 4] - v[b + 3]:6d}  barrier {v[b + 5] - v[b + 4]:6d}   step total {v[b + 5] - prev:6d}")
                print(f"  block total {v[3 + 6 * 7 + 5] - v[0]} cycles")


if __name__ == "__main__":
    main()
