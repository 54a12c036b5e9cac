#!/usr/bin/env python3
"""Golden fixtures of the reference's loss classes (src/loss/loss.py), produced by RUNNING THE REFERENCE on the CPU in the build
container.  The module imports torchvision at its top for the two losses that need it (sigmoid focal loss, TransformLoss); torchvision
is not installed here, so placeholder modules are registered first and only the classes that never touch it are exercised.  Nothing of
the reference is copied: inputs come from seeded generators below (the tests regenerate them), only outputs and gradients are stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_losses.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PASN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

for name in ("torchvision", "torchvision.ops", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.models"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["torchvision.ops"].sigmoid_focal_loss = None
sys.modules["torchvision.transforms.functional"].affine = None
sys.modules["torchvision.transforms.functional"].InterpolationMode = types.SimpleNamespace(BILINEAR="bilinear")

from src.loss import loss as ref  # noqa: E402  (reference)

sys.path.insert(0, os.path.join(REPO, "tests"))
from loss_cases import CASES, make_inputs  # noqa: E402  (shared with tests/test_cpu_losses.py)


def main():
    out = {}
    for tag, cls_name, kwargs, kind in CASES:
        args = make_inputs(kind)
        leaf = args[0].clone().requires_grad_()
        loss = getattr(ref, cls_name)(**kwargs).compute(leaf, *args[1:])
        loss.backward()
        out[tag + "_loss"] = loss.detach().numpy()
        out[tag + "_grad"] = leaf.grad.numpy()
    path = os.path.join(HERE, "g6_losses.npz")
    np.savez_compressed(path, **out)
    print(f"g6_losses.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(CASES)} cases")


if __name__ == "__main__":
    main()
