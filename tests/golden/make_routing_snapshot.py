"""Writes tests/golden/routing_x3d_s_cfg2.json: the DEFAULT launch list of the benchmarked configuration (BASELINE config 2: X3D-S,
32 x 3 x 16 x 224 x 224, bf16) -- one entry per launch: plan kind, kernel instance, layer shape.  Geometry only, runs without a GPU:

    python tests/golden/make_routing_snapshot.py

tests/test_cpu_routing.py compares the plan compiled with NO PASN_* switch set against this file, so an environment variable or a
refactor cannot silently change the benchmarked path; regenerate it when a routing change is intended (and say so in the commit)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def routing(arch="x3d_s", shape=(32, 3, 16, 224, 224)):
    import torch

    from protoasnet_amd import backbones, plan

    trunk = backbones.X3DFeatures(arch)
    pb = plan.PlanBuilder(torch.device("cpu"), torch.bfloat16, torch.bfloat16)
    x = pb.input(shape)
    with torch.no_grad():
        trunk.build_plan(pb, x)
    return [{"kind": m.get("kind", ""), "kernel": m["kernel"], "shape": m.get("shape", "")} for m in pb.meta]


if __name__ == "__main__":
    for k in [k for k in os.environ if k.startswith("PASN_")]:
        del os.environ[k]
    rows = routing()
    out = os.path.join(HERE, "routing_x3d_s_cfg2.json")
    with open(out, "w") as fh:
        json.dump({"workload": "x3d_s 32x3x16x224x224 bf16", "launches": len(rows), "rows": rows}, fh, indent=1)
    print(f"{out}: {len(rows)} launches")
