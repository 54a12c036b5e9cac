#!/usr/bin/env python3
"""Depthwise 3x3x3 weight gradient (pasn_dwconv3d_wgrad, bf16) at the X3D-S training shapes, one line per layer shape and arm.

    python tools/dwwg_bench.py "name=ENV=V ENV2=V" ...      (no arms: the default route against round 2's marching kernel)
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protoasnet_amd import _lib
from protoasnet_amd._lib import ConvDesc

DEV = torch.device("cuda")
# (channels, stride, T, H, W of the input)
SHAPES = [(54, 2, 16, 112, 112), (54, 1, 16, 56, 56), (108, 2, 16, 56, 56), (108, 1, 16, 28, 28), (216, 2, 16, 28, 28), (216, 1, 16, 14, 14),
          (432, 2, 16, 14, 14), (432, 1, 16, 7, 7)]


def main():
    arms = [a.split("=", 1) for a in sys.argv[1:]] or [["march2", ""], ["march", "PASN_DWWG_MARCH2=0"]]
    lib = _lib.lib()
    N = 32
    for c, s, t, h, w in SHAPES:
        cp = (c + 7) // 8 * 8
        ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
        d = ConvDesc(N=N, Ti=t, Hi=h, Wi=w, Cin=c, Cin_p=cp, To=t, Ho=ho, Wo=wo, Cout=c, Cout_p=cp, kt=3, kh=3, kw=3, st=1, sh=s, sw=s, pt=1, ph=1, pw=1)
        x = torch.randn(N, t, h, w, cp, device=DEV).bfloat16()
        dy = torch.randn(N, t, ho, wo, cp, device=DEV).bfloat16()
        mb = (x.numel() + dy.numel()) * 2 / 1e6
        line = f"{c:4d} ch s{s} {h:3d}x{w:<3d} {mb:7.1f} MB "
        ref = None
        for name, envs in arms:
            env = dict(e.split("=") for e in envs.split()) if envs else {}
            with _lib.tuning_env(**env):
                ws = torch.empty(int(lib.pasn_dwconv3d_wgrad_workspace_floats(ctypes.byref(d))), device=DEV)
                dw = torch.zeros(c, 27, device=DEV)
                run = lambda: _lib.check(lib.pasn_dwconv3d_wgrad(x.data_ptr(), dy.data_ptr(), ws.data_ptr(), dw.data_ptr(), ctypes.byref(d), 1, _lib.current_stream()))
                for _ in range(3):
                    run()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 100)
                if ref is None:
                    ref = dw.clone()
                err = float((dw - ref).abs().max() / ref.abs().max())
            line += f" | {name} {best:7.1f} us {mb / best * 1e-3 * 1e3:6.0f} GB/s (rel {err:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
