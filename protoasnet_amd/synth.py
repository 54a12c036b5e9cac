"""Deterministic synthetic inputs and parameters.

The reference ships no weights, fixtures or datasets that can travel (SURVEY.md
section 8c/8d), so every parity pin and every benchmark in this repo is driven by
the two recipes below.  They are plain numpy so the very same bytes can be fed to
the reference (when the golden fixtures are generated), to the CPU oracle and to
the HIP path.

* ``echo_clips`` follows the reference input contract: one grey channel in
  [0, 1), normalised with mean 0.099 / std 0.171 and repeated to three identical
  channels (reference ``src/data/as_dataloader.py:168-182``).
* ``synth_state_dict`` fills a ``state_dict`` by parameter *name and shape* only,
  so a reference module and a module of this package that expose the same keys
  receive identical values.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping

import numpy as np
import torch

ECHO_MEAN = 0.099  # src/data/as_dataloader.py:180
ECHO_STD = 0.171  # src/data/as_dataloader.py:181
DEFAULT_SEED = 200  # train.seed in every reference config (Ours_ProtoASNet_Video.yml:20)


def echo_clips(shape, seed: int = DEFAULT_SEED, dtype=torch.float32) -> torch.Tensor:
    """Synthetic echo batch.

    ``shape`` is (N, 3, T, H, W) for video or (N, 3, H, W) for images.  One grey
    channel is drawn and expanded to the three identical channels the reference
    dataloader produces.
    """
    shape = tuple(int(s) for s in shape)
    assert shape[1] == 3, "echo clips have 3 (identical) channels"
    grey_shape = (shape[0], 1) + shape[2:]
    rng = np.random.default_rng(seed)
    u = rng.random(grey_shape, dtype=np.float32)
    u = (u - np.float32(ECHO_MEAN)) / np.float32(ECHO_STD)
    x = torch.from_numpy(u).expand(*shape).contiguous()
    return x.to(dtype)


def echo_labels(n: int, num_real_classes: int = 3, seed: int = DEFAULT_SEED) -> torch.Tensor:
    """Integer AS labels in [0, num_real_classes) (as_dataloader.py:22 has 3 real classes)."""
    rng = np.random.default_rng(seed + 1)
    return torch.from_numpy(rng.integers(0, num_real_classes, size=(n,), dtype=np.int64))


def _rng_for(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode("utf-8"))])


def synth_tensor(name: str, shape, seed: int = DEFAULT_SEED) -> np.ndarray:
    """Value of one parameter / buffer, decided by its name suffix and shape."""
    shape = tuple(int(s) for s in shape)
    rng = _rng_for(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return rng.uniform(0.6, 1.4, size=shape).astype(np.float32)
    if leaf == "prototype_vectors":
        return rng.random(shape, dtype=np.float32)  # torch.rand in ProtoPNet.py:132
    if leaf == "ones":
        return np.ones(shape, dtype=np.float32)  # ProtoPNet.py:136
    if leaf == "bias":
        return (0.05 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == "weight" and len(shape) == 1:
        # norm-layer scale: below one so a deep residual trunk keeps O(1) activations
        return rng.uniform(0.5, 0.9, size=shape).astype(np.float32)
    if leaf == "weight" and len(shape) == 2:
        # last_layer (K, P): dense, signed
        return (0.5 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        std = np.sqrt(2.0 / max(fan_in, 1))
        if name.startswith("add_on_layers"):
            std *= 0.3  # keeps PPNet's trailing Sigmoid (ProtoPNet.py:129) out of saturation on O(5) trunk features
        return (std * rng.standard_normal(shape)).astype(np.float32)
    return (0.1 * rng.standard_normal(shape)).astype(np.float32)


def synth_state_dict(template: Mapping[str, torch.Tensor], seed: int = DEFAULT_SEED) -> Dict[str, torch.Tensor]:
    """A full ``state_dict`` for ``template`` (only names, shapes and dtypes are read)."""
    out = {}
    for name, t in template.items():
        v = torch.from_numpy(synth_tensor(name, tuple(t.shape), seed))
        out[name] = v.to(t.dtype) if t.dtype != v.dtype else v
    return out


def load_synth(module: torch.nn.Module, seed: int = DEFAULT_SEED) -> torch.nn.Module:
    """Overwrite every parameter and buffer of ``module`` with the recipe, in place."""
    module.load_state_dict(synth_state_dict(module.state_dict(), seed), strict=True)
    return module
