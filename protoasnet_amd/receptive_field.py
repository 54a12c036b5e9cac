"""Receptive-field bookkeeping the PPNet / XProtoNet constructors need (host-side, integers and halves).

Mirrors the interface of the reference's ``src/utils/receptive_field.py`` (``compute_proto_layer_rf_info_v2``
:109-134, ``compute_rf_prototype`` :61-66): ``proto_layer_rf_info = [n, j, r, start]`` with n the feature-map
side, j the jump (product of strides), r the receptive-field size and start the centre of the first field.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Union

Padding = Union[int, str]


def _step(rf: List[float], k: int, s: int, pad: Padding) -> List[float]:
    n, j, r, start = rf
    if pad == "SAME":
        n_out = math.ceil(n / s)
        total = max(k - s, 0) if n % s == 0 else max(k - (n % s), 0)
    elif pad == "VALID":
        n_out = math.ceil((n - k + 1) / s)
        total = 0
    else:
        total = 2 * int(pad)
        n_out = (n - k + total) // s + 1
    left = total // 2
    return [n_out, j * s, r + (k - 1) * j, start + ((k - 1) / 2 - left) * j]


def compute_proto_layer_rf_info_v2(img_size: int, layer_filter_sizes: Sequence[int], layer_strides: Sequence[int],
                                   layer_paddings: Sequence[Padding], prototype_kernel_size: int) -> List[float]:
    if not (len(layer_filter_sizes) == len(layer_strides) == len(layer_paddings)):
        raise AssertionError("conv_info lists must have equal length")
    rf: List[float] = [img_size, 1, 1, 0.5]
    for k, s, p in zip(layer_filter_sizes, layer_strides, layer_paddings):
        rf = _step(rf, k, s, p)
    return _step(rf, prototype_kernel_size, 1, "VALID")


def compute_rf_prototype(img_size: int, prototype_patch_index, protoL_rf_info) -> List[int]:
    """[image index, h0, h1, w0, w1] pixel box seen by the patch at (image, h, w)."""
    n, j, r, start = protoL_rf_info
    img, hi, wi = prototype_patch_index
    if not (hi < n and wi < n):
        raise AssertionError("patch index outside the prototype layer")
    box = [img]
    for idx in (hi, wi):
        centre = start + idx * j
        box += [max(int(centre - r / 2), 0), min(int(centre + r / 2), img_size)]
    return box
