#!/usr/bin/env python3
"""Headline benchmark: clips/sec forward of Video ProtoASNet on synthetic echo batches (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          # no launcher: this process starts N fresh rank processes itself (see spawn_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one eval-mode forward (logits, similarity, occurrence_map) of the whole model -- trunk + prototype layer --
over one batch of 32 clips of (3,16,224,224), inputs resident in HBM, weights random (deterministic recipe), bf16
activations with fp32 accumulation.  Multi-GPU: every rank runs its own batch (weak scaling, no data-path collective:
clips are independent units); timing = barrier + synchronize on both sides, MAX over ranks.

The JSON line also carries
  roofline     -- the dominant kernel instance (by device time): algorithmic bytes per launch / its average launch
                  duration, measured with HIP events recorded on the launch stream inside the timed region (around its launches in
                  three of the timed steps: first, middle, last -- bracketing every launch of every step cost the job 1.5-2 %);
  cpu_baseline -- the CPU oracle (restated reference op sequence, torch fp32, all host cores) on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: ~6.3 TB/s achievable; measured on this pool's boxes: copy 4.7-5.3 TB/s, fill 6.9 -- DESIGN.md section 6)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.0}  # dense peaks, same guide (never the 2:1-sparsity figures)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def spawn_ranks(gpus: int, script: str, argv) -> None:
    """Make ``--gpus N`` self-contained and honest.

    Under a launcher (WORLD_SIZE set) the world size must equal --gpus, otherwise the run is refused.  Without a launcher
    and N > 1, THIS process -- which has not touched the GPU (importing torch and counting devices do not) -- starts
    ``python -m torch.distributed.run`` with N fresh rank processes as a child, lets rank 0's JSON line through on stdout and
    exits with the child's return code (non-zero if any rank failed).  Never an exec after GPU initialisation."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            raise SystemExit(f"--gpus {gpus} but the launcher set WORLD_SIZE={ws}; start exactly one rank per requested GPU")
        return
    if gpus <= 1:
        return
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def init_ranks(args_gpus: int):
    """(world, rank, device, backend) of this rank; joins the process group when world > 1 and proves it with a collective."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args_gpus, f"world size {world} != --gpus {args_gpus}"
    backend = os.environ.get("PASN_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"--gpus {world} but only {ndev} GPU(s) are visible (RCCL needs one device per rank; "
                         "PASN_BENCH_BACKEND=gloo rehearses several ranks on one card)")
    dev = torch.device("cuda", local_rank % max(ndev, 1)) if ndev else torch.device("cpu")
    if ndev:
        torch.cuda.set_device(dev)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist

        # RCCL ("nccl") over xGMI on a real node.  PASN_BENCH_BACKEND=gloo rehearses the N > 1 code path with several
        # ranks on ONE GPU (RCCL refuses two ranks on the same device); the forward data path has no collective either way.
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        ones = torch.ones(1, dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)  # every rank contributes 1: the sum is the number of ranks that really joined
        ranks_seen = int(ones.item())
        assert ranks_seen == world, f"all-reduce saw {ranks_seen} ranks, expected {world}"
    return world, rank, dev, backend, ranks_seen


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # 0.27 s of device time; at 20 the fixed cost of the timed region (barrier, event bracketing of three steps) reads as 1.5 %
    ap.add_argument("--no-graph", action="store_true", help="enqueue every launch from the host instead of replaying the captured hipGraph")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--arch", default="x3d_s", choices=["x3d_s", "x3d_m", "resnet2p1d_18"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--prototypes", type=int, default=30)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--cpu-clips", type=int, default=96, help="max clips in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="stop the CPU-baseline sample after this much CPU work")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--per-op", default="", help="write a per-launch timing table to this file")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="forward = the BASELINE.json headline metric (default); train = config 3's training step (tools/train_bench.py)")
    ap.add_argument("--train-loss", default="reference", choices=["reference", "simple"], help="--mode train: the reference's loss recipe "
                    "(incl. TransformLoss's second trunk pass) or a single-pass loss")
    ap.add_argument("--input", default="clip3", choices=["clip3", "grey"], help="clip3 = the reference's (N,3,T,H,W) clip (the headline "
                    "metric); grey = the single-channel clip of the device-side input pipeline (protoasnet_amd.data): an extra, "
                    "labelled as such in config.workload")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs measured after the headline (R(2+1)D forward, "
                    "config 5, config 4 push sweep, config 3 train step: reported under the \"secondary\" key, N = 1 only)")
    ap.add_argument("--dry-run", action="store_true", help="launcher rehearsal for the CPU test suite: ranks, process group, "
                    "barriers, MAX-over-ranks timing and the JSON line, with NO device work (the metric says so)")
    return ap.parse_args()


def main():
    args = parse()
    spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if args.dry_run:
        return dry_run(args)
    if args.mode == "train":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import train_bench

        return train_bench.main(["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch),
                                 "--frames", str(args.frames), "--size", str(args.size), "--arch", args.arch, "--dtype", args.dtype,
                                 "--loss", args.train_loss]
                                + (["--per-op", args.per_op] if args.per_op else []))
    world, rank, dev, backend, ranks_seen = init_ranks(args.gpus)
    if dev.type != "cuda":
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback (use --dry-run to rehearse the launcher)")

    from protoasnet_amd import model_builder, synth

    cfg = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture=args.arch, backbone_last_layer_num=-3,
               pretrained=False, prototype_shape=f"({args.prototypes}, 256, 1, 1, 1)", num_classes=args.classes,
               img_size=args.size)
    model = model_builder.build(cfg)
    synth.load_synth(model)
    cpu_state = {k: v.clone() for k, v in model.state_dict().items()} if (rank == 0 and world == 1 and args.cpu_clips) else None
    model = model.to(dev).eval()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.set_compute_dtype(dtype)
    shape = (args.batch, 3, args.frames, args.size, args.size)
    x_cpu = synth.echo_clips(shape, seed=synth.DEFAULT_SEED + rank)
    x = x_cpu.to(dev).to(dtype)  # resident in HBM before the timed region
    if args.input == "grey":  # one (already normalised) channel: the first layer runs with weights summed over the input channels
        x = x[:, :1].contiguous()
    trunk = model.cnn_backbone

    # The step: hipGraph replay of the whole forward (protoasnet_amd/graph.py: the launch list is static per shape, captured once;
    # --no-graph: one host enqueue per launch).  Steps that bracket launches with HIP events (the roofline samples) run launch by launch.
    from protoasnet_amd.graph import GraphedForward
    graphed = None if args.no_graph else GraphedForward(model)
    if graphed is not None:
        try:
            graphed(x)  # capture now: a capture that fails is reported and the run goes on launch by launch (same kernels, same results)
        except Exception as e:  # noqa: BLE001
            print(f"bench: hipGraph capture failed ({e!r}); falling back to one host enqueue per launch", file=sys.stderr)
            graphed = None
            torch.cuda.synchronize()

    def step():
        if graphed is not None and getattr(trunk, "_timers", None) is None:
            return graphed(x)
        with torch.no_grad():
            return model(x)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (also compiles the plan) + pick the dominant kernel instance by device time -----------------
    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    plan = trunk.plan_for(x)
    dominant, timers = None, None
    if not args.no_roofline:
        probe = {i: [] for i in range(len(plan.ops))}
        trunk._timers = probe
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        trunk._timers = None
        # The dominant kernel is chosen by FAMILY (the __global__ function, all template instances together): per instance the step's
        # largest entries are within 2 % of one another (8 stride-1 Swish stencils ~ 2 fused expand + stencil launches) and the choice
        # flipped from run to run; by family the depthwise stencil (22 launches, ~28 % of the step) is the largest by a wide margin.
        per_kernel, per_instance = {}, {}
        for i, evs in probe.items():
            ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
            inst = plan.meta[i]["kernel"]
            fam = inst.split("<")[0]
            per_kernel.setdefault(fam, [0.0, []])
            per_kernel[fam][0] += ms
            per_kernel[fam][1].append(i)
            per_instance.setdefault(inst, [0.0, 0, 0.0, 0.0])
            pi = per_instance[inst]
            pi[0] += ms
            pi[1] += 1
            pi[2] += plan.meta[i]["bytes"]
            pi[3] += plan.meta[i]["flops"]
        if args.per_op and rank == 0:  # per-launch table for the optimisation log (not part of the JSON contract)
            with open(args.per_op, "w") as fh:
                for i, evs in probe.items():
                    ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
                    m = plan.meta[i]
                    fh.write(f"{i:3d} {m['kernel']:34s} {m.get('shape', ''):44s} {ms * 1e3:9.1f} us {m['bytes'] / 1e6:9.1f} MB "
                             f"{m['bytes'] / ms / 1e6:8.1f} GB/s {m['flops'] / ms / 1e9:8.1f} TF/s\n")
        dominant = max(per_kernel, key=lambda k: per_kernel[k][0])
        timers = {i: [] for i in per_kernel[dominant][1]}
        kernel_ms = {k: round(v[0], 4) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][0])}
    else:
        kernel_ms = {}

    # ---- timed region ------------------------------------------------------------------------------------------
    # Pure hipGraph replay: nothing but the step itself sits between the two barriers.
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # roofline samples: three more steps of the same workload AFTER the timed region, launch by launch with HIP events around the
    # dominant kernel's launches on the launch stream (round 5: inside the region they were 15 % of the driver's 20-step line)
    if timers is not None:
        trunk._timers = timers
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    trunk._timers = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    logits = out[0]
    assert torch.isfinite(logits).all(), "non-finite logits"

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    clips = args.batch * world * args.steps
    value = clips / elapsed
    result = {
        "metric": "clips/sec forward", "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "ranks_seen": ranks_seen,
        "collective_backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"Video ProtoASNet forward, {args.arch} trunk + prototype layer (P={args.prototypes}, D=256, "
                               f"K={args.classes}), {args.batch}x{args.frames}x{args.size}x{args.size} echo clips per GPU"
                               + (" [single-channel grey input: device-side input pipeline, NOT the headline configuration]" if args.input == "grey" else ""),
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (independent clips, no collective)",
                   "launch": "host enqueue per launch" if graphed is None else
                             ("hipGraph replay of the forward's launch list (one graph launch per step"
                              + ("" if args.no_roofline else "; the 3 roofline-sample steps run AFTER the timed region, launch by launch with HIP events "
                                                            "around the dominant kernel's launches") + ")"),
                   "tuning": _tuning_in_force()},
    }

    # ---- roofline of the dominant kernel -------------------------------------------------------------------------
    total_bytes = sum(m["bytes"] for m in plan.meta)
    total_flops = sum(m["flops"] for m in plan.meta)
    if dominant is not None:
        durs = [a.elapsed_time(b) for evs in timers.values() for a, b in evs]  # ms, one per launch in the timed region
        avg_ms = sum(durs) / len(durs)
        idx = list(timers.keys())
        avg_bytes = sum(plan.meta[i]["bytes"] for i in idx) / len(idx)
        avg_flops = sum(plan.meta[i]["flops"] for i in idx) / len(idx)
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), collected in
        # their own runs by tools/pmc_traffic.py -- a profiler cannot run inside this process
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                table = json.load(open(pmc_path))
                # launch-weighted mean over the family's instances as the profiler names them (it prints defaulted template arguments
                # the plan's names leave out)
                rows = [v for k, v in table.items() if k.split("<")[0] == dominant and v.get("launches_sampled")]
                if rows:
                    traffic = int(sum(v["traffic_bytes_per_launch"] * v["launches_sampled"] for v in rows) / sum(v["launches_sampled"] for v in rows))
                    traffic_source = ("profiles/pmc_traffic.json (STORED rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, run "
                                      f"{table.get('_run', {}).get('tag', 'untagged')}; not measured in this process)")
            except (ValueError, OSError):
                traffic = None
        tflops = avg_flops / (avg_ms * 1e-3) / 1e12
        # which roof bounds this kernel: its arithmetic intensity against the machine balance (dense bf16 MFMA peak / HBM peak)
        mfma_peak = MFMA_PEAK_TFLOPS[args.dtype]
        if avg_flops / max(avg_bytes, 1.0) > mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9):
            result["roofline"] = {
                "kernel": dominant, "bound": "mfma", "achieved": round(tflops, 1), "peak": mfma_peak, "unit": "TFLOP/s",
                "frac": round(tflops / mfma_peak, 4), "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": len(idx),
                "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": int(avg_bytes),
                "algorithmic_flops_per_launch": int(avg_flops), "hbm_gbs": round(achieved, 1),
            }
        else:
            result["roofline"] = {
                "kernel": dominant, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": len(idx),
                "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": int(avg_bytes),
                "mfma_tflops": round(tflops, 2),
            }
        # the launch-by-launch probe (an event pair around every launch) over-reads the graph-replayed step by a few per cent:
        # kernel_ms_per_step is the probe's split scaled to the timed step, the raw probe sum is kept beside it
        probe_sum = sum(kernel_ms.values())
        ratio = (1000.0 * elapsed / args.steps) / probe_sum if probe_sum > 0 else 1.0
        result["kernel_ms_per_step"] = {k: round(v * ratio, 4) for k, v in kernel_ms.items()}
        result["kernel_ms_probe"] = {"sum_ms": round(probe_sum, 4), "scaled_by": round(ratio, 4),
                                     "note": "per-family device time from HIP events around every launch of two untimed steps (trunk launches "
                                             "only; the head's two launches are inside ms_per_step), scaled so that the families sum to ms_per_step"}
        # the five largest template instances with their own fractions (event-timed in the two probe steps): bound by the same rule
        balance = MFMA_PEAK_TFLOPS[args.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)
        top = sorted(per_instance.items(), key=lambda kv: -kv[1][0])[:5]
        result["roofline_instances"] = [
            {"kernel": k, "launches_per_step": v[1], "ms_per_step": round(v[0], 4),
             **({"bound": "mfma", "frac": round(v[3] / (v[0] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 4)} if v[3] / max(v[2], 1.0) > balance
                else {"bound": "hbm", "frac": round(v[2] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})}
            for k, v in top]
    # Two byte counts for the whole step.  "bytes_per_clip" is what the launches of THIS build move algorithmically (a fused launch counts
    # its own inputs and outputs only).  "layerwise_bytes_per_clip" is SURVEY section 8(d)'s accounting -- every layer's input and output
    # once, norm / activation fused, residual read once -- i.e. the same plan compiled with every cross-layer fusion switched off; it does
    # not move when a fusion removes traffic, so it is the figure the throughput can be compared against across rounds.
    layerwise = _layerwise_bytes(trunk, x, dtype)
    result["trunk_algorithmic"] = {
        "bytes_per_clip": int(total_bytes / args.batch), "flops_per_clip": int(total_flops / args.batch),
        "hbm_frac_whole_step": round(total_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4),
        "layerwise_bytes_per_clip": int(layerwise / args.batch) if layerwise else None,
        "hbm_frac_whole_step_layerwise": round(layerwise * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4) if layerwise else None,
        "arena_bytes": plan.arena_bytes,
        "note": "bytes_per_clip counts what this build's launches move (fused launches skip their intermediates); layerwise_bytes_per_clip is the "
                "per-layer count of SURVEY 8(d), constant across builds: compare rounds on hbm_frac_whole_step_layerwise",
    }

    # ---- CPU baseline: the oracle (restated reference op sequence) on a bounded sample of the same workload ---------
    if cpu_state is not None:
        import oracle

        # a 1-GPU box owns a 16-core share of the host (the node reports all 256 hardware threads; using them
        # oversubscribes the share and runs ~10x slower), so the baseline uses at most 16 threads and says so
        cores = min(len(os.sched_getaffinity(0)), 16)
        torch.set_num_threads(cores)
        # bounded sample: the batch's clips re-used round robin, 8 at a time (bounds the broadcast-product intermediate),
        # until ~args.cpu_seconds of CPU work or args.cpu_clips clips are done
        chunk, done, sims, logit_refs = 8, 0, [], []
        with torch.no_grad():
            oracle.nets.xprotonet_forward(cpu_state, x_cpu[:1].float(), arch=args.arch)  # warm-up
            t0 = time.perf_counter()
            while done < args.cpu_clips and (time.perf_counter() - t0) < args.cpu_seconds:
                lo = done % args.batch
                xs = x_cpu[lo: lo + min(chunk, args.batch - lo)].float()
                ref = oracle.nets.xprotonet_forward(cpu_state, xs, arch=args.arch)
                if done < args.batch:
                    sims.append((lo, ref["similarity"]))
                    logit_refs.append((lo, ref["logits"]))
                done += xs.shape[0]
            dt = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": round(done / dt, 3), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{done} clips of 3x{args.frames}x{args.size}x{args.size} in chunks of {chunk}, fp32 torch oracle "
                      f"(reference op sequence incl. broadcast-product pooling), {dt:.1f} s of CPU work",
        }
        # the same clips through the HIP path must agree with what the CPU computed (bf16 tolerance)
        err = max(float((out[1][lo: lo + s.shape[0]].float().cpu() - s).abs().max()) for lo, s in sims)
        result["cpu_baseline"]["max_abs_similarity_diff_vs_gpu"] = round(err, 5)
        err_l = max(float((out[0][lo: lo + s.shape[0]].float().cpu() - s).abs().max()) for lo, s in logit_refs)
        result["parity_observed"] = {
            "clips_checked": sum(s.shape[0] for _, s in sims), "dtype": args.dtype, "against": "fp32 CPU oracle on the timed batch's own clips",
            "similarity_max_abs": round(err, 6), "logits_max_abs": round(err_l, 6),
            "gates": {"similarity": 1e-3, "logits": 2e-3, "where": "tests/test_gpu_models.py (BF16_SIM, BF16_LOGITS)"}}
    # ---- the other BASELINE configs, measured in this process AFTER the headline's timed region (the headline is untouched) ---------
    if world == 1 and not args.no_secondary and args.arch == "x3d_s" and args.input == "clip3":
        del model, trunk, plan, x, out
        torch.cuda.empty_cache()
        result["secondary"] = secondary(dev)
    print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


def _tuning_in_force():
    """PASN_* switches in the library's snapshot (csrc/tuning.h) + unknown PASN_* names in the environment: {} on a default run."""
    from protoasnet_amd import _lib

    out = {}
    for ln in _lib.tuning_report().splitlines():
        if ln.startswith("unknown: "):
            out["unknown"] = ln[9:].split()
        elif "=" in ln:
            k, v = ln.split("=", 1)
            out[k] = v
    return out


_UNFUSED = {"PASN_EXPDW": "0", "PASN_NO_XPAIR": "1", "PASN_WSPAIR": "0", "PASN_NO_SHORTFUSE": "1", "PASN_NO_SE_PROLOGUE": "1",
            "PASN_NO_SE_FUSE": "1"}


def _layerwise_bytes(trunk, x, dtype):
    """Algorithmic bytes of one batch through the trunk with every cross-layer fusion off (plan compiled, never run): SURVEY 8(d)'s
    per-layer accounting.  None for trunks without a plan builder."""
    from protoasnet_amd.plan import PlanBuilder

    if not hasattr(trunk, "build_plan"):
        return None
    from protoasnet_amd import _lib

    saved = {k: os.environ.get(k) for k in _UNFUSED}
    try:
        os.environ.update(_UNFUSED)
        _lib.tuning_reload()  # the library routes on ONE snapshot of the PASN_* switches (csrc/tuning.h)
        pb = PlanBuilder(x.device, dtype, dtype)
        x_in = pb.input(tuple(x.shape))
        pb.finish(x_in, trunk.build_plan(pb, x_in))
        return float(sum(m["bytes"] for m in pb.meta))
    except Exception as e:  # an accounting extra must never cost the bench line
        print(f"bench.py: layerwise byte count unavailable ({type(e).__name__}: {e})", file=sys.stderr)
        return None
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _lib.tuning_reload()


def _kernel_roofline(trunk, step, plan, dtype_name: str, reps: int = 3):
    """Dominant kernel instance of one forward (HIP events around every launch, ``reps`` passes) and its roofline entry."""
    probe = {i: [] for i in range(len(plan.ops))}
    trunk._timers = probe
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    trunk._timers = None
    per = {}
    for i, evs in probe.items():
        ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        e = per.setdefault(plan.meta[i]["kernel"], [0.0, 0.0, 0.0, 0])
        e[0] += ms
        e[1] += plan.meta[i]["bytes"]
        e[2] += plan.meta[i]["flops"]
        e[3] += 1
    name = max(per, key=lambda k: per[k][0])
    ms, nbytes, flops, n = per[name]
    gbs, tfl = nbytes / (ms * 1e-3) / 1e9, flops / (ms * 1e-3) / 1e12
    mfma_peak = MFMA_PEAK_TFLOPS[dtype_name]
    if flops / max(nbytes, 1.0) > mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9):
        roof = {"kernel": name, "bound": "mfma", "achieved": round(tfl, 1), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(tfl / mfma_peak, 4)}
    else:
        roof = {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    roof.update({"traffic": None, "launches_per_step": n, "avg_launch_us": round(ms / n * 1e3, 2), "share_of_step": round(ms / sum(v[0] for v in per.values()), 3)})
    return roof


def secondary(dev) -> dict:
    """BASELINE configs other than the headline, each a short measurement on this GPU (synthetic data, random-init weights, bf16):
    the reference's own video trunk (R(2+1)D-18[:-3], 8 x 32 x 112^2: reference-video shape of SURVEY section 8a) with its roofline,
    config 5 (X3D-M, 32 x 312^2, P = 60), config 4 (push sweep over 10 000 clips) and one config-3 training step with the reference's loss
    recipe.  Numbers only: parity of every one of these paths is the GPU test suite's job."""
    import random

    from protoasnet_amd import losses as L
    from protoasnet_amd import model_builder, synth
    from protoasnet_amd.push import push_prototypes

    bf16 = torch.bfloat16
    out = {}

    def build(arch, P, K, size):
        m = model_builder.build(dict(checkpoint_path="", name="Video_XProtoNet", base_architecture=arch, backbone_last_layer_num=-3,
                                     pretrained=False, prototype_shape=f"({P}, 256, 1, 1, 1)", num_classes=K, img_size=size))
        synth.load_synth(m)
        return m.to(dev)

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def guarded(key, fn):
        try:
            out[key] = fn()
        except Exception as e:  # a secondary line must never cost the headline its JSON
            out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    def r2p1d():
        m = build("resnet2p1d_18", 40, 4, 112).eval().set_compute_dtype(bf16)
        x = synth.echo_clips((8, 3, 32, 112, 112)).to(dev).to(bf16)

        def f():
            with torch.no_grad():
                return m(x)

        sec = timed(f, 4, 20)
        plan = m.cnn_backbone.plan_for(x)
        return {"workload": "Video ProtoASNet forward, R(2+1)D-18[:-3] trunk + prototype layer (P=40, D=256, K=4), 8x32x112x112 echo clips (the "
                            "reference's video configuration)", "value": round(8 / sec, 1), "unit": "clips/s", "ms_per_step": round(sec * 1e3, 3),
                "trunk_tflops": round(sum(mm["flops"] for mm in plan.meta) / sec / 1e12, 1), "roofline": _kernel_roofline(m.cnn_backbone, f, plan, "bf16")}

    def cfg5():
        # BASELINE config 5 / SURVEY 8(d): "fp32 vs bf16 tolerance sweep": clips/s in both precisions + the error table of bf16 against fp32
        m = build("x3d_m", 60, 3, 312).eval()
        x32 = synth.echo_clips((8, 3, 32, 312, 312)).to(dev)
        res = {"workload": "BASELINE config 5: X3D-M trunk + prototype layer (P=60, K=3), 8x32x312x312 echo clips, fp32 and bf16"}
        outs = {}
        for name, dt in (("fp32", torch.float32), ("bf16", bf16)):
            m.set_compute_dtype(dt)
            x = x32.to(dt)

            def f():
                with torch.no_grad():
                    return m(x)

            sec = timed(f, 2, 3 if name == "fp32" else 8)
            outs[name] = [t.float().clone() for t in f()]
            plan = m.cnn_backbone.plan_for(x)
            tb = sum(mm["bytes"] for mm in plan.meta)
            res[name] = {"value": round(8 / sec, 1), "unit": "clips/s", "ms_per_step": round(sec * 1e3, 3),
                         "hbm_frac_whole_step": round(tb / sec / 1e9 / HBM_PEAK_GBS, 4)}
            del x
        res["value"], res["unit"], res["ms_per_step"] = res["bf16"]["value"], "clips/s", res["bf16"]["ms_per_step"]
        res["hbm_frac_whole_step"] = res["bf16"]["hbm_frac_whole_step"]
        table = {}
        for key, a, b in zip(("logits", "similarity", "occurrence_map"), outs["fp32"], outs["bf16"]):
            d = (a - b).abs()
            table[key] = {"max_abs_err": round(float(d.max()), 6), "mean_abs_err": round(float(d.mean()), 7), "max_abs_fp32": round(float(a.abs().max()), 5),
                          "rel_mean_err": round(float(d.mean() / a.abs().mean().clamp_min(1e-30)), 6)}
        table["argmax_logit_agreement"] = float((outs["fp32"][0].argmax(1) == outs["bf16"][0].argmax(1)).float().mean())
        res["bf16_vs_fp32"] = table
        return res

    def cfg1():
        # BASELINE config 1 / BASELINE.md section 4 case (i): Ours_ProtoASNet_Image.yml:13-16,85 -- XProtoNet on ResNet-18, 10 prototypes per
        # class x 4 classes, 224^2 frames, batch 8 -- the configuration the reference itself can run on a CPU
        import oracle

        m = model_builder.build(dict(checkpoint_path="", name="XProtoNet", base_architecture="resnet18", pretrained=False,
                                     prototype_shape="(40, 512, 1, 1)", num_classes=4, img_size=224, add_on_layers_type="regular"))
        synth.load_synth(m)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        x = synth.echo_clips((8, 3, 224, 224))
        cores = min(len(os.sched_getaffinity(0)), 16)
        torch.set_num_threads(cores)
        with torch.no_grad():
            oracle.nets.xprotonet_forward(sd, x[:1], arch="resnet18")
            t0, reps = time.perf_counter(), 0
            while time.perf_counter() - t0 < 4.0:
                ref = oracle.nets.xprotonet_forward(sd, x, arch="resnet18")
                reps += 1
            t_cpu = (time.perf_counter() - t0) / reps
        m = m.to(dev).eval()
        res = {"workload": "BASELINE config 1: Image ProtoASNet (XProtoNet), ResNet-18, P=40, D=512, K=4, 8x3x224x224 synthetic frames",
               "unit": "images/s",
               "cpu_baseline": {"value": round(8 / t_cpu, 1), "unit": "images/s", "cores": cores, "kind": "port",
                                "sample": f"{reps} batches of 8 frames, fp32 torch oracle, {t_cpu * reps:.1f} s of CPU work"}}
        for name, dt in (("fp32", torch.float32), ("bf16", bf16)):
            m.set_compute_dtype(dt)
            xin = x.to(dev).to(dt)

            def f():
                with torch.no_grad():
                    return m(xin)

            sec = timed(f, 5, 50)
            out = f()
            res[name] = {"value": round(8 / sec, 1), "ms_per_step": round(sec * 1e3, 4),
                         "max_abs_similarity_diff_vs_cpu": round(float((out[1].float().cpu() - ref["similarity"]).abs().max()), 6)}
        res["value"], res["ms_per_step"] = res["bf16"]["value"], res["bf16"]["ms_per_step"]
        xb = synth.echo_clips((256, 3, 224, 224)).to(dev).to(bf16)

        def g():
            with torch.no_grad():
                return m(xb)

        res["bf16"]["images_per_s_batch256"] = round(256 / timed(g, 3, 10), 1)
        return res

    def cfg4():
        m = build("x3d_s", 30, 3, 224).eval().set_compute_dtype(bf16)
        xs = synth.echo_clips((32, 3, 16, 224, 224)).to(dev).to(bf16)

        class Loader:  # 10 000 clips = 312 batches of 32 + one of 16: the 32 resident clips re-labelled per batch
            batch_size = 32

            def __len__(self):
                return 313

            def __iter__(self):
                for b in range(313):
                    n = 32 if b < 312 else 10000 - 312 * 32
                    yield {"cine": xs[:n], "target_AS": (torch.arange(n) + b) % 3, "filename": None}

        with torch.no_grad():
            m(xs)  # compile the launch list outside the timed sweep
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = push_prototypes(Loader(), m, class_specific=True, abstain_class=False, replace_prototypes=True, log=lambda *_: None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        line = {"workload": "BASELINE config 4: push_prototypes sweep, 10 000 clips 3x16x224x224 (313 batches), X3D-S, 30 prototypes, class specific",
                "value": round(10000 / dt, 1), "unit": "clips/s", "seconds": round(dt, 3), "winners_found": int((res["proto_index"] >= 0).sum())}
        # BASELINE.md section 4 case (iii): the reference's push sweep on the host cores -- push_forward per batch (oracle), everything to numpy,
        # the per-prototype masked-argmin loop (push_abs_revision.py:268-307; its own timing is :213,347-348) -- on a bounded sample
        import numpy as np

        import oracle

        cores = min(len(os.sched_getaffinity(0)), 16)
        torch.set_num_threads(cores)
        sd = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
        xc = xs[:8].float().cpu()
        ident = m.prototype_class_identity.numpy()
        batches, t0, done = [], time.perf_counter(), 0
        with torch.no_grad():
            while done < 24 and time.perf_counter() - t0 < 10.0:
                ref = oracle.nets.xprotonet_forward(sd, xc, arch="x3d_s")
                occ = ref["occurrence_map"].numpy()  # the reference copies the whole map and the batch to the host as well (:278-285)
                batches.append((ref["features_extracted"].numpy(), 1.0 - ref["similarity"].numpy(), (np.arange(8) + done) % 3))
                done += 8
            oracle.push.xproto_push_select(batches, ident, 3, True, False)
        dtc = time.perf_counter() - t0
        del occ
        line["cpu_baseline"] = {"value": round(done / dtc, 3), "unit": "clips/s", "cores": cores, "kind": "port",
                                "sample": f"{done} clips (batches of 8) through the oracle's push_forward + the reference's numpy selection loop, {dtc:.1f} s of CPU work"}
        return line

    def cfg3():
        m = build("x3d_s", 30, 3, 224).train().set_compute_dtype(bf16)
        x = synth.echo_clips((32, 3, 16, 224, 224)).to(dev).to(bf16)
        labels = torch.randint(0, 3, (32,), generator=torch.Generator().manual_seed(0)).to(dev)
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
        ce, cluster = L.CeLoss(loss_weight=1, reduction="mean"), L.ClusterRoiFeat(loss_weight=0.8, num_classes=3, reduction="mean")
        separation = L.SeparationRoiFeat(loss_weight=0.08, num_classes=3, reduction="mean", abstain_class=False)
        trans = L.TransformLoss(loss_weight=1e-3, reduction="mean")
        fc_l1 = L.L_norm(mask=1 - torch.t(m.prototype_class_identity), p=1, loss_weight=1e-4)
        random.seed(1234)

        def step(paired=True):
            opt.zero_grad(set_to_none=True)
            if paired:  # the transform term's trunk pass over the warped clips rides in the forward's launch list (two statistics groups)
                (logits, sim, occ), t_loss = trans.paired_forward(x, m)
            else:       # ... or runs as the reference runs it: a second pass, model.compute_occurence_map (loss.py:302)
                logits, sim, occ = m(x)
                t_loss = trans.compute(x, occ, m)
            loss = (ce.compute(logits, labels) + cluster.compute(sim, labels) + separation.compute(sim, labels)
                    + t_loss + fc_l1.compute(m.last_layer.weight))
            loss.backward()
            opt.step()
            return loss

        sec = timed(step, 3, 10)  # (four timed steps read 46-48 ms for the same build; ten: +-0.3)
        sec_two = timed(lambda: step(False), 2, 5)
        line = {"workload": "BASELINE config 3, per-GPU work: one training step (forward + the reference's loss recipe incl. the transform term's "
                            "second trunk pass + backward + Adam), X3D-S + prototype layer, 32x16x224x224", "value": round(32 / sec, 1),
                "unit": "clips/s", "ms_per_step": round(sec * 1e3, 2),
                "passes": "ONE trunk pass over [clips, warped clips] with two batch-statistics groups (model.forward_pair): the 64 clips of the reference's "
                          "two passes, each half normalised with its own statistics",
                "ms_per_step_two_passes": round(sec_two * 1e3, 2)}
        # roofline of the C-ABI entry point that takes the most device time: one more step with every launch of the first-pass plan bracketed by
        # HIP events on the launch stream (tools/train_bench.py's accounting: bytes = the buffers a launch touches, each once)
        runner = next(r for r in m._train_runners.values() if r.mode == 2)  # the paired pass (64 clips, two statistics groups)
        plan, evs = runner.plan, []
        orig = list(plan.ops)

        def wrap(i, op):
            def run(ptrs, st):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                op(ptrs, st)
                b.record()
                evs.append((i, a, b))
            return run

        plan.ops[:] = [wrap(i, op) for i, op in enumerate(orig)]
        was_serial, plan.serial = plan.serial, True  # per-launch times: every launch on the bracketed stream (the timed steps above overlap the weight gradients)
        try:
            step()
            torch.cuda.synchronize()
        finally:
            plan.ops[:] = orig
            plan.serial = was_serial
        agg = {}
        for i, a, b in evs:
            k = ("fwd " if i < plan.n_fwd else "bwd ") + plan.op_names[i]
            e = agg.setdefault(k, [0.0, 0, 0])
            e[0] += a.elapsed_time(b)
            e[1] += 1
            e[2] += plan.op_bytes[i]
        top, (ms, n, nb) = max(agg.items(), key=lambda kv: kv[1][0])
        line["roofline"] = {"kernel": top, "bound": "hbm", "achieved": round(nb / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(nb / ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": None, "launches_per_step": n, "avg_launch_us": round(1e3 * ms / n, 2),
                            "algorithmic_bytes_per_launch": int(nb / n), "share_of_pass": round(ms / sum(v[0] for v in agg.values()), 3),
                            "note": "the paired trunk pass of the step (launch list of forward_pair(): forward + backward over 64 clips); bytes = buffers touched, each once"}
        line["device_ms_by_entry_point"] = {k: round(v[0], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]}
        return line

    guarded("r2plus1d_18_forward", r2p1d)
    guarded("config1_image_forward", cfg1)
    guarded("config5_x3d_m_forward", cfg5)
    guarded("config4_push_sweep", cfg4)
    guarded("config3_train_step", cfg3)
    return out


def dry_run(args):
    """Launcher rehearsal (CPU test suite): everything around the step -- ranks, process group, barrier + MAX timing, JSON -- and
    a step that does no device work.  Its line cannot be mistaken for a measurement: metric and data say "dry-run"."""
    os.environ.setdefault("PASN_BENCH_BACKEND", "gloo")
    world, rank, dev, backend, ranks_seen = init_ranks(args.gpus)
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "dry-run (launcher rehearsal, no device work)", "value": 0.0, "unit": "clips/s", "n_gpus": world,
                          "ranks_seen": ranks_seen, "collective_backend": backend if world > 1 else None, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "dry-run",
                          "config": {"workload": "none", "global_batch": args.batch * world}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
