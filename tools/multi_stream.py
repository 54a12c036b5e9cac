"""Experiment: does splitting the 32-clip batch over S HIP streams (S independent plans of 32/S clips, replayed concurrently) hide the
per-launch fixed costs (ramp, weight loads, tail) that bound the small layers?   python tools/multi_stream.py --streams 1 2 4"""
import argparse
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    from protoasnet_amd import model_builder, synth

    dev = torch.device("cuda:0")
    cfg = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture="x3d_s", backbone_last_layer_num=-3, pretrained=False,
               prototype_shape="(40, 256, 1, 1, 1)", num_classes=4, img_size=224)
    base = model_builder.build(cfg)
    synth.load_synth(base)
    x_all = synth.echo_clips((args.batch, 3, 16, 224, 224), seed=synth.DEFAULT_SEED).to(dev).to(torch.bfloat16)
    ref = None
    for rnd in range(args.rounds):
        for S in args.streams:
            models = [copy.deepcopy(base).to(dev).eval() for _ in range(S)]
            for m in models:
                m.set_compute_dtype(torch.bfloat16)
            xs = [c.contiguous() for c in x_all.chunk(S)]
            streams = [torch.cuda.Stream() for _ in range(S)]

            def step():
                outs = []
                with torch.no_grad():
                    for m, xx, st in zip(models, xs, streams):
                        with torch.cuda.stream(st):
                            outs.append(m(xx)[0])
                return outs

            for _ in range(5):
                outs = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                outs = step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            logits = torch.cat([o.float() for o in outs])
            if ref is None:
                ref = logits
            print(f"round {rnd} streams={S}: {dt * 1e3:7.3f} ms/step  {args.batch / dt:8.1f} clips/s   max|dlogits| vs first {float((logits - ref).abs().max()):.3g}", flush=True)
            del models


if __name__ == "__main__":
    main()
