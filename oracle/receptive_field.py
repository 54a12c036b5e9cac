"""Oracle receptive-field arithmetic (TEST INFRASTRUCTURE -- see oracle/__init__.py).

The PPNet / XProtoNet constructors store ``proto_layer_rf_info = [n, j, r, start]``
(src/models/ProtoPNet.py:353-360); resnet18 at 224 px gives ``[7, 32, 435, 0.5]``.
"""
from __future__ import annotations

import math


def layer_rf(filter_size, stride, padding, prev):
    """One layer of the recurrence -- src/utils/receptive_field.py:4-33 (integer-padding and
    'SAME' / 'VALID' branches)."""
    n_in, j_in, r_in, start_in = prev
    if padding == "SAME":
        n_out = math.ceil(float(n_in) / float(stride))
        pad = max(filter_size - stride, 0) if n_in % stride == 0 else max(filter_size - (n_in % stride), 0)
    elif padding == "VALID":
        n_out = math.ceil(float(n_in - filter_size + 1) / float(stride))
        pad = 0
    else:
        pad = padding * 2
        n_out = math.floor((n_in - filter_size + pad) / stride) + 1
    pL = math.floor(pad / 2)
    return [n_out, j_in * stride, r_in + (filter_size - 1) * j_in, start_in + ((filter_size - 1) / 2 - pL) * j_in]


def proto_layer_rf_info_v2(img_size, filter_sizes, strides, paddings, prototype_kernel_size):
    """src/utils/receptive_field.py:109-134."""
    assert len(filter_sizes) == len(strides) == len(paddings)
    rf = [img_size, 1, 1, 0.5]
    for f, s, p in zip(filter_sizes, strides, paddings):
        rf = layer_rf(f, s, p, rf)
    return layer_rf(prototype_kernel_size, 1, "VALID", rf)


def rf_prototype(img_size, patch_index, rf_info):
    """src/utils/receptive_field.py:36-66 -- [img, h0, h1, w0, w1] box of one prototype patch."""
    n, j, r, start = rf_info
    img, hi, wi = patch_index
    assert hi < n and wi < n
    ch, cw = start + hi * j, start + wi * j
    return [
        img,
        max(int(ch - r / 2), 0),
        min(int(ch + r / 2), img_size),
        max(int(cw - r / 2), 0),
        min(int(cw + r / 2), img_size),
    ]
