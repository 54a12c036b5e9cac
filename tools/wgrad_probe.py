#!/usr/bin/env python3
"""Time one windowed weight gradient through the C-ABI (optimisation tool):  python tools/wgrad_probe.py 64 144 133 8 32 56 56"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoasnet_amd import _lib
from protoasnet_amd._lib import ConvDesc

cin, cout, kind, n, t, h, w = [int(v) for v in sys.argv[1:8]]
k, p = ((1, 3, 3), (0, 1, 1)) if kind == 133 else ((3, 1, 1), (1, 0, 0))
rup = lambda v, m: (v + m - 1) // m * m
d = ConvDesc(N=n, Ti=t, Hi=h, Wi=w, Cin=cin, Cin_p=rup(cin, 8), To=t, Ho=h, Wo=w, Cout=cout, Cout_p=rup(cout, 8), kt=k[0], kh=k[1], kw=k[2],
             st=1, sh=1, sw=1, pt=p[0], ph=p[1], pw=p[2])
lib = _lib.lib()
dev = torch.device("cuda")
x = torch.randn(n, t, h, w, d.Cin_p, device=dev).bfloat16()
dy = torch.randn(n, t, h, w, d.Cout_p, device=dev).bfloat16()
taps = k[0] * k[1] * k[2]
dw = torch.zeros(cout, cin, taps, device=dev)
nb = int(lib.pasn_conv3d_wgrad_workspace_bytes(ctypes.byref(d), 1))
ws = torch.empty(max(nb, 4) // 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(lib.pasn_conv3d_wgrad_ws(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ctypes.byref(d), 1, ws.data_ptr() if nb else 0, st))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"wgrad {cin}->{cout} k{kind} {n}x{t}x{h}x{w}: {e0.elapsed_time(e1) * 100:.1f} us  ws {nb / 1e6:.1f} MB  env={ {k_: v for k_, v in os.environ.items() if k_.startswith('PASN_')} }")
