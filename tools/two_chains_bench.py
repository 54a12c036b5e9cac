#!/usr/bin/env python3
"""Experiment: the 32-clip forward as C concurrent chains of 32 / C clips on C streams inside one hipGraph (kernels of different chains
overlap each other's ramp and tail) against the one-chain graph.  python tools/two_chains_bench.py [steps]"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protoasnet_amd import model_builder, synth
from protoasnet_amd.graph import GraphedForward

DEV = torch.device("cuda")


def build():
    cfg = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture="x3d_s", backbone_last_layer_num=-3, pretrained=False,
               prototype_shape="(30, 256, 1, 1, 1)", num_classes=3, img_size=224)
    m = model_builder.build(cfg)
    synth.load_synth(m)
    m = m.to(DEV).eval()
    m.set_compute_dtype(torch.bfloat16)
    return m


class Chains(torch.nn.Module):
    def __init__(self, model, chains):
        super().__init__()
        self.models = torch.nn.ModuleList([model] + [copy.deepcopy(model) for _ in range(chains - 1)])
        self.streams = [torch.cuda.Stream() for _ in range(chains - 1)]

    def forward(self, x):
        c = len(self.models)
        parts = x.chunk(c, dim=0)
        cur = torch.cuda.current_stream()
        outs = [None] * c
        for s in self.streams:
            s.wait_stream(cur)
        outs[0] = self.models[0](parts[0])
        for i, s in enumerate(self.streams):
            with torch.cuda.stream(s):
                outs[i + 1] = self.models[i + 1](parts[i + 1])
        for s in self.streams:
            cur.wait_stream(s)
        return outs


def time_it(fn, steps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps)
    return best


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    model = build()
    x = synth.echo_clips((32, 3, 16, 224, 224)).to(DEV).bfloat16()
    g1 = GraphedForward(model)
    ref = [t.clone() for t in g1(x)]
    t1 = time_it(lambda: g1(x), steps)
    print(f"1 chain : {t1:.4f} ms  {32 / t1 * 1e3:.0f} clips/s", flush=True)
    for c in (2, 4):
        ch = Chains(model, c).eval()
        ch.training = False
        gc = GraphedForward(ch)
        outs = gc(x)
        torch.cuda.synchronize()
        # same results as the one-chain forward (per-clip work; SE pool partial rows may be summed in another order)
        for k in range(len(ref)):
            got = torch.cat([o[k] for o in outs], dim=0) if ref[k].shape[0] == 32 else None
            if got is not None:
                print(f"   output {k}: max |diff| vs one chain {float((got.float() - ref[k].float()).abs().max()):.3e}")
        tc = time_it(lambda: gc(x), steps)
        print(f"{c} chains: {tc:.4f} ms  {32 / tc * 1e3:.0f} clips/s", flush=True)


if __name__ == "__main__":
    main()
