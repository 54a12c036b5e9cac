#!/usr/bin/env python3
"""A gated (squeeze-excite) project + expand pair of the 14 x 14 stage against the same pair without a gate, per launch (HIP events around each
launch of the plan): where do the ~14 us of a gate go?

    python tools/se_pair_bench.py
    PASN_LIB_PATH=.../libprotoasnet_amd_tuning.so PASN_WS_ABL=64 python tools/se_pair_bench.py      # gate rows of ones instead of the FC chain
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from protoasnet_amd.plan import Act, PlanBuilder

DEV = torch.device("cuda")


def build(se, N=32, T=16, hw=14, cm=216, c=96, cse=16):
    torch.manual_seed(3)
    pb = PlanBuilder(DEV, torch.bfloat16, torch.bfloat16)
    x = torch.relu(torch.randn(N, T, hw, hw, cm, device=DEV)).bfloat16()
    r = torch.relu(torch.randn(N, T, hw, hw, c, device=DEV)).bfloat16()
    xa = Act(N, T, hw, hw, cm, cm, pb._new_buf(x.numel() * 2, external=True))
    ra = Act(N, T, hw, hw, c, c, pb._new_buf(r.numel() * 2, external=True))
    m = [nn.Conv3d(cm, cm, 3, 1, 1, groups=cm, bias=False), nn.BatchNorm3d(cm), nn.Conv3d(cm, c, 1, bias=False), nn.BatchNorm3d(c),
         nn.Conv3d(c, cm, 1, bias=False), nn.BatchNorm3d(cm), nn.Conv3d(cm, cse, 1), nn.Conv3d(cse, cm, 1)]
    m = [mm.to(DEV).eval() for mm in m]
    if se:
        y, pooled = pb.dwconv(xa, m[0], m[1], act="none", pool=True)
        o1, o2 = pb.conv_pair(y, m[2], m[3], "relu", ra, m[4], m[5], "relu", in_swish=True, se=(pooled, m[6], m[7]))
    else:
        y = pb.dwconv(xa, m[0], m[1], act="swish")
        o1, o2 = pb.conv_pair(y, m[2], m[3], "relu", ra, m[4], m[5], "relu")
    plan = pb.finish(xa, o2)
    plan.ptrs[ra.buf] = r.data_ptr()
    return plan, x, [mm["kernel"] for mm in pb.meta]


def main():
    for se in (False, True):
        plan, x, names = build(se)
        for _ in range(3):
            plan.run(x)
        torch.cuda.synchronize()
        timers = {i: [] for i in range(len(plan.ops))}
        for _ in range(20):
            plan.run(x, timers)
        torch.cuda.synchronize()
        line = f"se={se}:"
        for i, nm in enumerate(names):
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in timers[i])
            line += f"  {nm} {ts[len(ts) // 2]:6.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
