#!/usr/bin/env python3
"""Pointwise (1x1x1) weight gradient (pasn_conv3d_wgrad, bf16) at the wide X3D-S training shapes, one line per shape and arm.

    python tools/pwwg_bench.py [N] "name=ENV=V ..." ...
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protoasnet_amd import _lib
from protoasnet_amd._lib import ConvDesc

DEV = torch.device("cuda")
SHAPES = [(96, 216, 14), (216, 96, 14), (192, 432, 7), (432, 192, 7), (48, 108, 28), (108, 48, 28), (48, 216, 28), (96, 432, 14)]


def main():
    args = sys.argv[1:]
    N = int(args.pop(0)) if args and args[0].isdigit() else 64
    arms = [a.split("=", 1) for a in args] or [["xcd", ""], ["grid2d", "PASN_WGT_XCD=0"]]
    lib = _lib.lib()
    for ci, co, hw in SHAPES:
        cip, cop = (ci + 7) // 8 * 8, (co + 7) // 8 * 8
        d = ConvDesc(N=N, Ti=16, Hi=hw, Wi=hw, Cin=ci, Cin_p=cip, To=16, Ho=hw, Wo=hw, Cout=co, Cout_p=cop, kt=1, kh=1, kw=1, st=1, sh=1, sw=1, pt=0, ph=0, pw=0)
        x = torch.randn(N, 16, hw, hw, cip, device=DEV).bfloat16()
        dy = torch.randn(N, 16, hw, hw, cop, device=DEV).bfloat16()
        mb = (x.numel() + dy.numel()) * 2 / 1e6
        line = f"{ci:4d} -> {co:4d} {hw:2d}x{hw:<2d} N={N} {mb:7.1f} MB "
        ref = None
        for name, envs in arms:
            env = dict(e.split("=") for e in envs.split()) if envs else {}
            with _lib.tuning_env(**env):
                dw = torch.zeros(co, ci, device=DEV)

                def run():
                    dw.zero_()
                    _lib.check(lib.pasn_conv3d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ctypes.byref(d), 1, _lib.current_stream()))

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
            line += f" | {name} {best:7.1f} us {mb / best:5.2f} TB/s (rel {err:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
