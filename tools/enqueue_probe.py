#!/usr/bin/env python3
"""Host-side enqueue time of one forward step vs its device time (is the bench loop launch-bound anywhere?).
    python tools/enqueue_probe.py            # X3D-S, 32 x 16 x 224 x 224 bf16"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoasnet_amd import model_builder, synth

dev = torch.device("cuda")
cfg = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture="x3d_s", backbone_last_layer_num=-3, pretrained=False,
           prototype_shape="(30, 256, 1, 1, 1)", num_classes=3, img_size=224)
m = model_builder.build(cfg)
synth.load_synth(m)
m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
x = synth.echo_clips((32, 3, 16, 224, 224)).to(dev).to(torch.bfloat16)
with torch.no_grad():
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        m(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"MAXW={os.environ.get('PASN_DWMFMA_MAXW', 'default')}: enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, total {1e3 * (t2 - t0) / K:.3f} ms/step")
