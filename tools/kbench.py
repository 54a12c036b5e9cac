#!/usr/bin/env python3
"""Micro-benchmark single trunk launches through the C-ABI (optimisation tool, not part of the driver contract).

    python tools/kbench.py pw 24 54 16 112 112          # pointwise conv  Cin Cout T H W   (N = 32, bf16)
    python tools/kbench.py pwres 54 24 16 56 56         # ... with residual
    python tools/kbench.py dw 54 1 16 56 56             # depthwise 3x3x3  C stride T H W
    python tools/kbench.py c133 64 144 32 56 56 8       # dense (1,3,3) conv Cin Cout T H W [N]  (c311: (3,1,1)) -- R(2+1)D layers
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from protoasnet_amd.plan import Act, PlanBuilder, round_up

DEV = torch.device("cuda")


def run(kind, a, b, T, H, W, N=32, dtype=torch.bfloat16, reps=20):
    pb = PlanBuilder(DEV, dtype, dtype)
    es = 2 if dtype == torch.bfloat16 else 4
    cin = a
    cp = round_up(cin, 8)
    x = torch.randn(N, T, H, W, cp, device=DEV).to(dtype)
    xa = Act(N, T, H, W, cin, cp, pb._new_buf(x.numel() * es, external=True))
    res_t = None
    if kind in ("pw", "pwres"):
        conv = nn.Conv3d(cin, b, 1, bias=False).to(DEV)
        bn = nn.BatchNorm3d(b).to(DEV).eval()
        ra = None
        if kind == "pwres":
            res_t = torch.randn(N, T, H, W, round_up(b, 8), device=DEV).to(dtype)
            ra = Act(N, T, H, W, b, round_up(b, 8), pb._new_buf(res_t.numel() * es, external=True))
        y = pb.conv(xa, conv, bn, "relu", residual=ra)
    elif kind in ("c133", "c311"):
        k, p = ((1, 3, 3), (0, 1, 1)) if kind == "c133" else ((3, 1, 1), (1, 0, 0))
        conv = nn.Conv3d(cin, b, k, 1, p, bias=False).to(DEV)
        y = pb.conv(xa, conv, nn.BatchNorm3d(b).to(DEV).eval(), "relu")
        ra = None
    else:
        conv = nn.Conv3d(cin, cin, 3, (1, b, b), 1, groups=cin, bias=False).to(DEV)
        bn = nn.BatchNorm3d(cin).to(DEV).eval()
        y = pb.dwconv(xa, conv, bn, os.environ.get("PASN_KB_ACT", "none"))  # PASN_KB_ACT=swish: the non-SE blocks' epilogue
        ra = None
    plan = pb.finish(xa, y)
    if ra is not None:
        plan.ptrs[ra.buf] = res_t.data_ptr()
    for _ in range(3):
        plan.run(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.run(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    m = plan.meta[0]
    print(f"{kind} {a}->{b} {T}x{H}x{W} N={N} {m['kernel']:36s} {ms*1e3:8.1f} us  {m['bytes']/1e6:8.1f} MB  {m['bytes']/ms/1e6:7.0f} GB/s"
          f"  env={ {k: v for k, v in os.environ.items() if k.startswith('PASN_')} }")


if __name__ == "__main__":
    kind = sys.argv[1]
    a, b, T, H, W = (int(v) for v in sys.argv[2:7])
    run(kind, a, b, T, H, W, N=int(sys.argv[7]) if len(sys.argv) > 7 else 32)
