#!/usr/bin/env python3
"""Would two half-batch chains on two streams beat one full-batch chain?  (measurement probe, not the product path)"""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protoasnet_amd import model_builder, synth

dev = torch.device("cuda")
cfg = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture="x3d_s", backbone_last_layer_num=-3,
           pretrained=False, prototype_shape="(30, 256, 1, 1, 1)", num_classes=3, img_size=224)
def make():
    m = model_builder.build(cfg)
    synth.load_synth(m)
    m = m.to(dev).eval()
    m.set_compute_dtype(torch.bfloat16)
    return m
ma, mb = make(), make()
x = synth.echo_clips((32, 3, 16, 224, 224)).to(dev).to(torch.bfloat16)
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
xs = [c.contiguous() for c in x.chunk(parts)]
trunks = [ma.cnn_backbone, mb.cnn_backbone] + [make().cnn_backbone for _ in range(parts - 2)]
streams = [torch.cuda.Stream() for _ in range(parts)]

def full():
    return ma.cnn_backbone(x)

def split():
    cur = torch.cuda.current_stream()
    outs = []
    for s, t, xi in zip(streams, trunks, xs):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(t(xi))
    for s in streams:
        cur.wait_stream(s)
    return outs

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
    print(f"trunk, one chain of 32 clips : {timeit(full):.3f} ms")
    print(f"trunk, {parts} chains on {parts} streams: {timeit(split):.3f} ms")
