#!/usr/bin/env python3
"""Training-step benchmark (BASELINE.json config 3): clips/sec of forward + loss + backward + gradient all-reduce + Adam step
of Video ProtoASNet (X3D-S trunk, head B) on synthetic echo batches, one process per GPU.

    python tools/train_bench.py --steps 10 --warmup 3 [--dtype bf16|f32] [--batch 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/train_bench.py --gpus N

Loss (``--loss reference``, default): the recipe of src/configs/Ours_ProtoASNet_Video.yml:31-58 as the agent applies it
(Video_XProtoNet_e2e.py:88-100) -- CeLoss (mean) + 0.8 ClusterRoiFeat + 0.08 SeparationRoiFeat + 1e-3 TransformLoss (a SECOND trunk
pass, with gradients, over the affinely warped clip: model.compute_occurence_map) + 1e-4 L1 of the last layer's off-class weights;
``--loss simple`` keeps one pass (cross entropy + L1 of the maps + a cluster term).  protoasnet_amd.losses supplies the classes.
The optimizer is torch.optim.Adam (lr 1e-4) as in the reference's agents (Video_XProtoNet_e2e.py:36-62)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--arch", default="x3d_s")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--per-op", default="", help="write per-launch device times of one step to this file")
    ap.add_argument("--loss", default="reference", choices=["reference", "reference_two_passes", "simple"])
    args = ap.parse_args(argv)
    import bench  # repo root: the rank launcher / process-group set-up shared with the forward benchmark

    if argv is None:  # stand-alone: --gpus N > 1 without a launcher starts its own ranks (bench.py does this for --mode train)
        bench.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    world, rank, dev, backend, ranks_seen = bench.init_ranks(args.gpus)
    if dev.type != "cuda":
        raise SystemExit("train_bench.py needs a GPU: the HIP path has no CPU fallback")
    import torch.distributed as dist

    from protoasnet_amd import dp, model_builder, synth

    cfg = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture=args.arch, backbone_last_layer_num=-3, pretrained=False,
               prototype_shape="(30, 256, 1, 1, 1)", num_classes=3, img_size=args.size)
    model = model_builder.build(cfg)
    synth.load_synth(model)
    model = model.to(dev).train()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.set_compute_dtype(dtype)
    x = synth.echo_clips((args.batch, 3, args.frames, args.size, args.size), seed=synth.DEFAULT_SEED + rank).to(dev).to(dtype)
    labels = torch.randint(0, 3, (args.batch,), generator=torch.Generator().manual_seed(rank)).to(dev)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    ident = model.prototype_class_identity.to(dev)
    from protoasnet_amd import losses as L

    ce, cluster = L.CeLoss(loss_weight=1, reduction="mean"), L.ClusterRoiFeat(loss_weight=0.8, num_classes=3, reduction="mean")
    separation = L.SeparationRoiFeat(loss_weight=0.08, num_classes=3, reduction="mean", abstain_class=False)
    trans = L.TransformLoss(loss_weight=1e-3, reduction="mean")
    fc_l1 = L.L_norm(mask=1 - torch.t(model.prototype_class_identity), p=1, loss_weight=1e-4)
    import random

    random.seed(1234 + rank)

    def step():
        opt.zero_grad(set_to_none=True)
        if args.loss == "reference":  # the transform term's second trunk pass in the forward's launch list (model.forward_pair)
            (logits, sim, occ), t_loss = trans.paired_forward(x, model)
            loss = (ce.compute(logits, labels) + cluster.compute(sim, labels) + separation.compute(sim, labels)
                    + t_loss + fc_l1.compute(model.last_layer.weight))
        elif args.loss == "reference_two_passes":  # ... as a second pass (model.compute_occurence_map, loss.py:302)
            logits, sim, occ = model(x)
            loss = (ce.compute(logits, labels) + cluster.compute(sim, labels) + separation.compute(sim, labels)
                    + trans.compute(x, occ, model) + fc_l1.compute(model.last_layer.weight))
        else:
            logits, sim, occ = model(x)
            own = ident[:, labels].t()  # (N, P): prototypes of the clip's class
            loss = F.cross_entropy(logits, labels) + 1e-3 * occ.abs().mean() + 0.1 * ((1 - sim) * own).sum(1).mean()
        loss.backward()
        dp.allreduce_gradients(params)
        opt.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(loss), "non-finite loss"
    exchange = _time_exchange(params, world, dev) if world > 1 else None
    if rank == 0 and os.environ.get("PASN_TB_TORCHPROF"):
        # which torch-side ops (weight repacking, gradient accumulation, optimizer) a step issues, with their Python call sites
        from torch.profiler import ProfilerActivity, profile

        with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
            step()
            step()
        with open(os.environ["PASN_TB_TORCHPROF"], "w") as fh:
            fh.write(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cpu_time_total", row_limit=60, max_name_column_width=40))
    runner = next(r for r in model._train_runners.values() if r.mode == (2 if args.loss == 'reference' else 0))
    roofline = per_entry = None
    if rank == 0:
        # one more step with every launch of the first-pass plan bracketed by HIP events on the launch stream: device time per C-ABI entry
        # point, and the roofline of the entry point that takes the most time (bytes = the buffers its launches touch, each buffer once)
        plan = runner.plan
        evs = []
        orig = list(plan.ops)
        def wrap(i, op):
            def run(ptrs, st):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); op(ptrs, st); b.record(); evs.append((i, a, b))
            return run
        plan.ops[:] = [wrap(i, op) for i, op in enumerate(orig)]
        was_serial, plan.serial = plan.serial, True  # per-launch times: every launch on the bracketed stream
    if world > 1 or rank == 0:
        step(); torch.cuda.synchronize()  # EVERY rank runs the bracketed step: it contains the gradient exchange (a collective)
    if rank == 0:
        plan.ops[:] = orig
        plan.serial = was_serial
        agg = {}
        for i, a, b in evs:
            k = ('fwd ' if i < plan.n_fwd else 'bwd ') + plan.op_names[i]
            e = agg.setdefault(k, [0.0, 0, 0])
            e[0] += a.elapsed_time(b); e[1] += 1; e[2] += plan.op_bytes[i]
        per_entry = {k: {"ms": round(ms, 3), "launches": n, "GB/s": round(nb / ms / 1e6, 1) if ms > 0 else None}
                     for k, (ms, n, nb) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]}
        top, (ms, n, nb) = max(agg.items(), key=lambda kv: kv[1][0])
        roofline = {"kernel": top, "bound": "hbm", "achieved": round(nb / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(nb / ms / 1e6 / 8000.0, 4), "traffic": None, "launches_per_step": n, "avg_launch_us": round(1e3 * ms / n, 2),
                    "algorithmic_bytes_per_launch": int(nb / n), "note": "first-pass plan of the step; bytes = buffers touched, each once"}
        if args.per_op:
            with open(args.per_op, "w") as fh:
                tot = 0.0
                for i, a, b in evs:
                    ms_ = a.elapsed_time(b); tot += ms_
                    fh.write(f"{i:4d} {'fwd' if i < plan.n_fwd else 'bwd'} {plan.op_names[i]:32s} {ms_ * 1e3:9.1f} us {plan.op_bytes[i] / 1e6:9.1f} MB "
                             f"{plan.op_bytes[i] / max(ms_, 1e-6) / 1e6:8.0f} GB/s\n")
                fh.write(f"total {tot:.3f} ms over {len(evs)} launches (event-bracketed, includes launch gaps)\n")
                for k, (ms_, n_, nb_) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                    fh.write(f"SUM {k:40s} {n_:4d} launches {ms_:9.3f} ms {nb_ / 1e6:10.1f} MB {nb_ / max(ms_, 1e-6) / 1e6:8.0f} GB/s\n")
    if rank == 0:
        plan = runner.plan
        print(json.dumps({
            "metric": "clips/sec train step", "value": round(args.batch * world * args.steps / elapsed, 2), "unit": "clips/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "collective_backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Video ProtoASNet train step (fwd + {'reference loss recipe incl. TransformLoss second pass' if args.loss == 'reference' else 'simple loss'}"
                                   f" + bwd + grad all-reduce + Adam), {args.arch} + head B, "
                                   f"{args.batch}x{args.frames}x{args.size}x{args.size} echo clips per GPU",
                       "global_batch": args.batch * world, "parallelism": f"dp{world} (one flat fp32 gradient bucket all-reduced per step)"},
            "launches": {"forward": plan.n_fwd, "backward": len(plan.ops) - plan.n_fwd}, "arena_bytes": plan.arena_bytes,
            "naive_bytes": plan.naive_bytes, "grad_bucket_bytes": plan.gsize * 4, "loss": round(float(loss.detach()), 5),
            "max_memory_allocated_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
            "roofline": roofline, "device_ms_by_entry_point": per_entry, "exchange": exchange,
        }))
    if world > 1:
        dist.destroy_process_group()


def _time_exchange(params, world, dev, reps: int = 10):
    """The step's one collective measured ALONE (one-shot vs ring), after the timed region, on the gradients as the last step
    left them: bucket bytes, and device time per exchange (HIP events on the launch stream, MAX over ranks) of three formulations --
    ``all_reduce`` (what the step uses: RCCL picks ring / tree), ``reduce_scatter + all_gather`` (the ring's two halves as explicit calls)
    and ``all_gather + local sum`` (one-shot: every rank pulls every peer's whole bucket over its own xGMI links, world x the bytes, one
    step).  Gradients are restored afterwards.  Only meaningful on RCCL; on gloo (CPU rehearsal) the numbers are host times."""
    import torch.distributed as dist

    from protoasnet_amd import dp

    grads = [p.grad for p in params if p.grad is not None]
    flat = dp.flat_gradient_view(grads)
    in_place = flat is not None
    if flat is None:
        flat = torch.cat([g.flatten() for g in grads])
    saved = flat.clone()
    gloo = dist.get_backend() == "gloo"
    n = flat.numel()
    pad = (-n) % world
    buf = torch.zeros(n + pad, dtype=flat.dtype, device=flat.device)
    gathered = torch.empty(world * (n + pad), dtype=flat.dtype, device=flat.device)
    shard = torch.empty((n + pad) // world, dtype=flat.dtype, device=flat.device)

    def timed(fn):
        try:
            fn()
            torch.cuda.synchronize()
            dist.barrier()
            if gloo:
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                ms = (time.perf_counter() - t0) * 1e3 / reps
            else:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    fn()
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / reps
            t = torch.tensor([ms], dtype=torch.float64, device="cpu" if gloo else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return round(float(t.item()), 4)
        except Exception as e:  # noqa: BLE001 -- a formulation the backend lacks must not cost the line
            return f"{type(e).__name__}: {e}"[:120]

    def h(t):  # gloo rehearsal: collectives on host copies
        return t.cpu() if gloo else t

    def f_allreduce():
        dist.all_reduce(h(buf))

    def f_rs_ag():
        dist.reduce_scatter_tensor(h(shard), h(buf))
        dist.all_gather_into_tensor(h(buf), h(shard))

    def f_oneshot():
        g = h(gathered)
        dist.all_gather_into_tensor(g, h(buf))
        g.view(world, -1).sum(0, out=h(buf))

    buf[:n].copy_(saved)
    out = {"bucket_bytes": n * flat.element_size(), "in_place_on_the_arena": in_place, "reps": reps, "unit": "ms per exchange (max over ranks)",
           "all_reduce": timed(f_allreduce), "reduce_scatter_all_gather": timed(f_rs_ag), "all_gather_local_sum_one_shot": timed(f_oneshot)}
    if isinstance(out["all_reduce"], float) and out["all_reduce"] > 0:
        out["all_reduce_bus_GBps"] = round(2 * (world - 1) / world * out["bucket_bytes"] / out["all_reduce"] / 1e6, 1)
    flat.copy_(saved)
    return out


if __name__ == "__main__":
    main()
