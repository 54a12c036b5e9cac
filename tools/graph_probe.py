#!/usr/bin/env python3
"""How much would hipGraph replay of the whole forward buy?  (measurement probe, not part of the product path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoasnet_amd import model_builder, synth

dev = torch.device("cuda")
cfg = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture="x3d_s", backbone_last_layer_num=-3,
           pretrained=False, prototype_shape="(30, 256, 1, 1, 1)", num_classes=3, img_size=224)
model = model_builder.build(cfg)
synth.load_synth(model)
model = model.to(dev).eval()
model.set_compute_dtype(torch.bfloat16)
x = synth.echo_clips((32, 3, 16, 224, 224)).to(dev).to(torch.bfloat16)

def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    eager = timeit(lambda: model(x))
    print(f"eager   {eager:.3f} ms/step")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = model(x)
        rep = timeit(g.replay)
        print(f"graph   {rep:.3f} ms/step")
    except Exception as e:  # noqa: BLE001
        print("capture failed:", repr(e)[:300])
