#!/usr/bin/env python3
"""The stride-1 depthwise stencil of the 14 x 14 stage (dw_tz.hip, Toeplitz form, against dwmfma.hip, block-diagonal form) at the X3D-S
benchmark shape (N = 32, T = 16, 216 channels, bf16), with and without squeeze-excite partial sums.

    python tools/dwtz_bench.py [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from protoasnet_amd import _lib
from protoasnet_amd.plan import Act, PlanBuilder

DEV = torch.device("cuda")


ENV = {"tz": dict(PASN_DW_TZ=None), "mfma": dict(PASN_DW_TZ="0")}


def build(arm, se, c, hw, N=32, T=16):
    with _lib.tuning_env(**ENV[arm]):
        torch.manual_seed(3)
        pb = PlanBuilder(DEV, torch.bfloat16, torch.bfloat16)
        cp = (c + 7) // 8 * 8
        x = torch.relu(torch.randn(N, T, hw, hw, cp, device=DEV)).bfloat16()
        x[..., c:] = 0
        xa = Act(N, T, hw, hw, c, cp, pb._new_buf(x.numel() * 2, external=True))
        m = [nn.Conv3d(c, c, 3, 1, 1, groups=c, bias=False).to(DEV).eval(), nn.BatchNorm3d(c).to(DEV).eval()]
        out = pb.dwconv(xa, m[0], m[1], "none" if se else "swish", pool=se)
        y = out[0] if se else out
        if se:
            pb.bufs[out[1][0]].external = True
        plan = pb.finish(xa, y)
        if se:
            pool_t = torch.zeros(N, out[1][1], y.Cp, dtype=torch.float32, device=DEV)
            plan.ptrs[out[1][0]] = pool_t.data_ptr()
            plan._keep = pool_t
        names = [mm["kernel"] for mm in pb.meta]
    return plan, x, names


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for c, hw in ((216, 14),):
        for se in (False, True):
            arms = {a: build(a, se, c, hw) for a in ("tz", "mfma")}
            times = {a: [] for a in arms}
            for _ in range(6):  # (the route is taken at launch: each arm runs under its own switches; first round = warm-up)
                for a in arms:
                    plan, x = arms[a][0], arms[a][1]
                    with _lib.tuning_env(**ENV[a]):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        plan.run(x)
                        e0.record()
                        for _ in range(reps):
                            plan.run(x)
                        e1.record()
                        torch.cuda.synchronize()
                        times[a].append(e0.elapsed_time(e1) * 1e3 / reps)
            print(f"{c} ch {hw}x{hw} se={se}: " + "   ".join(f"{a} {min(times[a]):6.1f} us ({' + '.join(arms[a][2])})" for a in arms), flush=True)
            if os.environ.get("PASN_TZ_STAMPS"):
                import ctypes
                h = ctypes.CDLL(_lib.LIB_PATH)
                buf = (ctypes.c_longlong * 62)()
                with _lib.tuning_env(**ENV["tz"]):
                    arms["tz"][0].run(arms["tz"][1])
                    torch.cuda.synchronize()
                h.pasn_debug_dt_stamps(buf)
                v = list(buf)
                print(f"  block {int(os.environ['PASN_TZ_STAMPS']) - 1} / wave 0 (shader cycles): operands {v[1] - v[0]}  prologue {v[2] - v[1]}")
                for k in range(8):
                    b = 3 + 5 * k
                    prev = v[2] if k == 0 else v[b - 1]
                    print(f"  step {k}: stencil {v[b] - prev:6d}  rows landed {v[b + 1] - v[b]:6d}  barrier + stores {v[b + 2] - v[b + 1]:6d}  pair transposed, next requested {v[b + 3] - v[b + 2]:6d}  barrier {v[b + 4] - v[b + 3]:6d}   step total {v[b + 4] - prev:6d}")
                print(f"  block total {v[3 + 5 * 7 + 4] - v[0]} cycles")


if __name__ == "__main__":
    main()
