#!/usr/bin/env python3
"""Prototype-layer kernels on their own (optimisation / evidence tool): head A (pasn_l2_head_fwd) and head B (pasn_xproto_head_fwd)
through the C-ABI on synthetic trunk features, at the shapes SURVEY section 8a names.  Prints one JSON line per case with the
algorithmic bytes / flops of SURVEY section 8d, the mean launch-sequence time (HIP events on the launch stream) and both rooflines.

    python tools/head_bench.py [--reps 50]
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protoasnet_amd import _lib, model_builder, synth
from protoasnet_amd.plan import round_up

DEV = torch.device("cuda")
HBM, MFMA = 8000.0, 2500.0


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def head_b(tag, N, Cb, spatial, P, K, D, dtype, reps, video=True):
    cfg = dict(checkpoint_path="", name="Video_XProtoNet" if video else "XProtoNet", base_architecture="x3d_s" if video else "resnet18",
               pretrained=False, prototype_shape=f"({P}, {D}, 1, 1{', 1' if video else ''})", num_classes=K, img_size=224)
    if video:
        cfg["backbone_last_layer_num"] = -3
    else:
        cfg["add_on_layers_type"] = "regular"
    m = model_builder.build(cfg)
    # swap the head's input width: the head kernels only see (Cb, D, P); rebuild the two first convs for Cb
    conv = torch.nn.Conv3d if video else torch.nn.Conv2d
    m.add_on_layers[0] = conv(Cb, D, 1)
    m.occurrence_module[0] = conv(Cb, D, 1)
    m = m.to(DEV).eval().set_compute_dtype(dtype)
    S = 1
    for v in spatial:
        S *= v
    feat_cl = torch.randn((N,) + tuple(spatial) + (round_up(Cb, 8),), device=DEV).clamp_(min=0).to(dtype)
    feat = feat_cl[..., :Cb].permute(0, len(spatial) + 1, *range(1, len(spatial) + 1))  # logical (N,C,...) view of channels-last rows

    class Trunk(torch.nn.Module):
        def forward(self, x):
            return feat

    m.cnn_backbone = Trunk()
    x = torch.zeros(1, device=DEV)
    with torch.no_grad():
        us = timed(lambda: m(x), reps)
    es = 2 if dtype == torch.bfloat16 else 4
    nbytes = N * (Cb * S * es + P * S * 4) + (P + K) * 4
    flops = N * (2 * S * (Cb * D + D * D + Cb * D + D * D // 2 + D // 2 * P + P * D) + 6 * P * D + 2 * P * K)
    print(json.dumps({"case": tag, "head": "B", "N": N, "Cb": Cb, "S": S, "P": P, "D": D, "dtype": str(dtype).split(".")[-1], "us": round(us, 1),
                      "algorithmic_MB": round(nbytes / 1e6, 2), "GFLOP": round(flops / 1e9, 2), "GB/s": round(nbytes / us / 1e3, 1),
                      "hbm_frac": round(nbytes / us / 1e3 / HBM, 4), "TFLOP/s": round(flops / us / 1e6, 1),
                      "mfma_frac": round(flops / us / 1e6 / MFMA, 4)}))


def head_a(tag, N, S, D, P, K, dtype, reps, want_dist=False):
    lib = _lib.lib()
    dp = round_up(D, 8)
    z = torch.rand(N, S, dp, device=DEV).to(dtype)
    protos = torch.rand(P, D, device=DEV)
    fcw = torch.randn(K, P, device=DEV)
    dist = torch.empty(N, P, S, device=DEV) if want_dist else None
    mind, amin, logits = torch.empty(N, P, device=DEV), torch.empty(N, P, dtype=torch.int32, device=DEV), torch.empty(N, K, device=DEV)

    def run():
        _lib.check(lib.pasn_l2_head_fwd(z.data_ptr(), protos.data_ptr(), fcw.data_ptr(), _lib.ptr(dist), mind.data_ptr(), amin.data_ptr(),
                                        logits.data_ptr(), N, S, D, dp, P, K, _lib.dtype_code(dtype), 0, 1e-4, _lib.current_stream()))

    us = timed(run, reps)
    es = 2 if dtype == torch.bfloat16 else 4
    nbytes = N * (D * S * es + P * 8 + (P * S * 4 if want_dist else 0)) + P * D * 4
    flops = N * (2 * P * D * S + 2 * D * S + 8 * P * S)
    print(json.dumps({"case": tag, "head": "A", "N": N, "S": S, "D": D, "P": P, "dtype": str(dtype).split(".")[-1], "dist_map": want_dist,
                      "us": round(us, 1), "algorithmic_MB": round(nbytes / 1e6, 3), "GFLOP": round(flops / 1e9, 3),
                      "GB/s": round(nbytes / us / 1e3, 1), "hbm_frac": round(nbytes / us / 1e3 / HBM, 4), "TFLOP/s": round(flops / us / 1e6, 2)}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    bf, f32 = torch.bfloat16, torch.float32
    head_a("cfg1 image (7x7, D=512, P=40)", 8, 49, 512, 40, 4, f32, a.reps)
    head_a("cfg1 image, bf16", 8, 49, 512, 40, 4, bf, a.reps)
    head_a("PPNet P=30, batch 256, bf16", 256, 49, 512, 30, 3, bf, a.reps)
    head_a("PPNet batch 256 + distance map (push_forward)", 256, 49, 512, 30, 3, bf, a.reps, want_dist=True)
    head_a("14x14 map (448 px), batch 64", 64, 196, 512, 30, 3, bf, a.reps)
    head_b("cfg2 X3D-S (Cb=192, S=784, P=30)", 32, 192, (16, 7, 7), 30, 3, 256, bf, a.reps)
    head_b("ref video (Cb=256, S=1568, P=40), N=5 fp32", 5, 256, (8, 14, 14), 40, 4, 256, f32, a.reps)
    head_b("ref video, N=32 bf16", 32, 256, (8, 14, 14), 40, 4, 256, bf, a.reps)
    head_b("cfg1 image head B (Cb=512, S=49, P=40, D=512), N=8 fp32", 8, 512, (7, 7), 40, 4, 512, f32, a.reps, video=False)
    head_b("cfg5 X3D-M (S=3200, P=60), N=8 bf16", 8, 192, (32, 10, 10), 60, 3, 256, bf, a.reps)
