#!/usr/bin/env python3
"""Measurements for the BASELINE.json configs that are not the headline bench line (run on the GPU box).  Lives under tests/ because
two of its legs use the oracle as the checker / CPU reference (tolerance table of config 5, CPU rate of config 1).

    python tests/configs_bench.py sweep     # config 5: X3D-M, 32x312x312 clips, 60 prototypes -- fp32 vs bf16 vs CPU oracle
    python tests/configs_bench.py push      # config 4: push_prototypes over 10k synthetic clips, 30 prototypes, 1 GPU
    python tests/configs_bench.py r2p1d     # reference-faithful trunk: R(2+1)D-18[:-3], 32x112x112 and 16x224x224 clips
    python tests/configs_bench.py image     # config 1: Image ProtoASNet (XProtoNet, ResNet-18, 40 prototypes, 224^2), batch 8: CPU oracle vs HIP
    python tests/configs_bench.py train     # config 3 at N = 1: training step of every trunk (fwd + loss + bwd + Adam), bf16
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protoasnet_amd import model_builder, synth

DEV = torch.device("cuda")


def build(arch, P, K, size):
    cfg = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture=arch, backbone_last_layer_num=-3,
               pretrained=False, prototype_shape=f"({P}, 256, 1, 1, 1)", num_classes=K, img_size=size)
    m = model_builder.build(cfg)
    synth.load_synth(m)
    return m


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def sweep():
    """fp32 HIP vs bf16 HIP vs fp32 CPU oracle on the same clips and weights (tolerance table of config 5)."""
    import oracle

    m = build("x3d_m", 60, 3, 312)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = synth.echo_clips((2, 3, 32, 312, 312))
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    t0 = time.perf_counter()
    ref = oracle.nets.xprotonet_forward(sd, x, arch="x3d_m", contract=True)  # (N,P,D,S) product would be 2x60x256x3200 floats
    t_cpu = time.perf_counter() - t0
    m = m.to(DEV).eval()
    rows = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m.set_compute_dtype(dt)
        with torch.no_grad():
            logits, sim, occ = m(x.to(DEV))
        torch.cuda.synchronize()
        rows[name] = {
            "max_abs_logits": float((logits.cpu() - ref["logits"]).abs().max()),
            "max_abs_similarity": float((sim.cpu() - ref["similarity"]).abs().max()),
            "max_abs_proto_dist": float(((1 - sim).cpu() - ref["proto_dist"]).abs().max()),
            "occ_mean_rel": float((occ.cpu() - ref["occurrence_map"]).abs().mean() / ref["occurrence_map"].abs().mean()),
        }
    # throughput at batch 8 (activations of one 32x312x312 clip are 3.9x those of a 16x224x224 clip)
    xb = synth.echo_clips((8, 3, 32, 312, 312)).to(DEV)
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m.set_compute_dtype(dt)
        xin = xb.to(dt)

        def f():
            with torch.no_grad():
                return m(xin)

        sec, _ = timed(f, 5)
        rows[name]["clips_per_s_batch8"] = round(8 / sec, 1)
    print(json.dumps({"config": "X3D-M 32x312x312, P=60, K=3, N=2 (tolerance) / N=8 (throughput)", "oracle_cpu_seconds_2clips": round(t_cpu, 1),
                      **rows}))


def push():
    """10 000 synthetic clips (313 batches of 32; the 32 resident clips are re-labelled per batch), class-specific mask on."""
    from protoasnet_amd.push import push_prototypes

    m = build("x3d_s", 30, 3, 224).to(DEV).eval().set_compute_dtype(torch.bfloat16)
    xs = synth.echo_clips((32, 3, 16, 224, 224)).to(DEV).to(torch.bfloat16)

    class Loader:
        batch_size = 32

        def __len__(self):
            return 313

        def __iter__(self):
            for b in range(313):
                n = 32 if b < 312 else 10000 - 312 * 32
                yield {"cine": xs[:n], "target_AS": (torch.arange(n) + b) % 3, "filename": None}

    t0 = time.perf_counter()
    out = push_prototypes(Loader(), m, class_specific=True, abstain_class=False, replace_prototypes=True, log=lambda *_: None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "push_prototypes, 10000 clips 3x16x224x224 bf16, X3D-S, 30 prototypes, class specific",
                      "seconds": round(dt, 3), "clips_per_s": round(10000 / dt, 1),
                      "winners_found": int((out["proto_index"] >= 0).sum())}))


def r2p1d():
    for shape, tag in (((8, 3, 32, 112, 112), "reference video config shape"), ((8, 3, 16, 224, 224), "BASELINE cfg-2 shape")):
        m = build("resnet2p1d_18", 40, 4, shape[-1]).to(DEV).eval().set_compute_dtype(torch.bfloat16)
        x = synth.echo_clips(shape).to(DEV).to(torch.bfloat16)

        def f():
            with torch.no_grad():
                return m(x)

        sec, _ = timed(f, 3)
        gmac = 76.02 if shape[2] == 32 else 152.04
        print(json.dumps({"config": f"Video ProtoASNet, R(2+1)D-18[:-3], {shape} bf16 ({tag})", "clips_per_s": round(shape[0] / sec, 1),
                          "ms_per_batch": round(sec * 1e3, 2), "trunk_TFLOPs": round(2 * gmac * shape[0] / sec / 1e3, 1)}))


def image():
    """Config 1 (Ours_ProtoASNet_Image.yml: ResNet-18, P=40, D=512, K=4, 224^2, batch 8) -- the reference's CPU-runnable case."""
    import oracle

    cfg = dict(checkpoint_path="", name="XProtoNet", base_architecture="resnet18", pretrained=False, prototype_shape="(40, 512, 1, 1)",
               num_classes=4, img_size=224, add_on_layers_type="regular")
    m = model_builder.build(cfg)
    synth.load_synth(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = synth.echo_clips((8, 3, 224, 224))
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    with torch.no_grad():
        oracle.nets.xprotonet_forward(sd, x[:1], arch="resnet18")
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 8.0:
            ref = oracle.nets.xprotonet_forward(sd, x, arch="resnet18")
            reps += 1
        t_cpu = (time.perf_counter() - t0) / reps
    m = m.to(DEV).eval()
    row = {"config": "Image ProtoASNet (XProtoNet), ResNet-18, P=40, D=512, K=4, 8x3x224x224 (BASELINE config 1)",
           "cpu_oracle_images_per_s": round(8 / t_cpu, 1), "cpu_threads": cores}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m.set_compute_dtype(dt)
        xin = x.to(DEV).to(dt)

        def f():
            with torch.no_grad():
                return m(xin)

        sec, out = timed(f, 50)
        row[name] = {"images_per_s_batch8": round(8 / sec, 1), "ms_per_batch": round(sec * 1e3, 3),
                     "max_abs_similarity_vs_cpu": float((out[1].float().cpu() - ref["similarity"]).abs().max())}
    xb = synth.echo_clips((256, 3, 224, 224)).to(DEV).to(torch.bfloat16)

    def g():
        with torch.no_grad():
            return m(xb)

    sec, _ = timed(g, 10)
    row["bf16"]["images_per_s_batch256"] = round(256 / sec, 1)
    print(json.dumps(row))


def train():
    """Training step (config 3's per-GPU work) for each trunk: forward + loss + backward + Adam, bf16 activations."""
    import torch.nn.functional as F

    for arch, shape, name in (("x3d_s", (32, 3, 16, 224, 224), "Video_XProtoNet"), ("x3d_m", (8, 3, 32, 312, 312), "Video_XProtoNet"),
                              ("resnet2p1d_18", (8, 3, 32, 112, 112), "Video_XProtoNet"), ("resnet18", (64, 3, 224, 224), "XProtoNet")):
        video = len(shape) == 5
        cfg = dict(checkpoint_path="", name=name, base_architecture=arch, pretrained=False, num_classes=3, img_size=shape[-1],
                   prototype_shape="(30, 256, 1, 1, 1)" if video else "(30, 512, 1, 1)")
        if video:
            cfg["backbone_last_layer_num"] = -3
        else:
            cfg["add_on_layers_type"] = "regular"
        m = model_builder.build(cfg)
        synth.load_synth(m)
        m = m.to(DEV).train().set_compute_dtype(torch.bfloat16)
        x = synth.echo_clips(shape).to(DEV).to(torch.bfloat16)
        labels = torch.randint(0, 3, (shape[0],), generator=torch.Generator().manual_seed(0)).to(DEV)
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-3)

        def step():
            opt.zero_grad(set_to_none=True)
            logits, sim, occ = m(x)
            loss = F.cross_entropy(logits, labels) + 1e-3 * occ.abs().mean() - 0.1 * sim.max(dim=1)[0].mean()
            loss.backward()
            opt.step()
            return loss

        sec, loss = timed(step, 5)
        plan = next(iter(m._train_runners.values())).plan
        print(json.dumps({"config": f"train step, {name} / {arch}, {shape} bf16", "clips_per_s": round(shape[0] / sec, 1),
                          "ms_per_step": round(sec * 1e3, 2), "launches": len(plan.ops), "arena_GB": round(plan.arena_bytes / 1e9, 2),
                          "loss": round(float(loss.detach()), 4)}))
        del m, opt, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    {"sweep": sweep, "push": push, "r2p1d": r2p1d, "image": image, "train": train}[sys.argv[1]]()
