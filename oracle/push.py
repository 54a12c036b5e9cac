"""Oracle prototype-push selection loops (TEST INFRASTRUCTURE -- see oracle/__init__.py).

numpy restatements of the *selection* part of the two reference push routines.
The plotting / pickling / receptive-field box code around them is out of scope
(SURVEY.md section 2.1 rows 8, 9).  The two routines break ties differently and
that difference is part of the contract:

* XProtoNet / Video push: a later batch wins a tie (``<=``), first index inside a batch.
* PPNet push: the first batch wins a tie (strict ``<``), first flattened (n_c, h, w) inside a batch.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np


def xproto_push_select(
    batches: Iterable[Tuple[np.ndarray, np.ndarray, np.ndarray]],
    prototype_class_identity: np.ndarray,
    num_classes: int,
    class_specific: bool = True,
    abstain_class: bool = True,
):
    """Selection loop of ``push_prototypes`` for XProtoNet / Video_XProtoNet.

    Reference: src/utils/push_abs_revision.py:226-237 (class masks; the last P/num_classes
    prototypes are not class specific when ``abstain_class``), :288-307 (per-batch loop).

    ``batches`` yields ``(protoL_input (B,P,D), proto_dist (B,P), gt (B,))`` in loader order.
    Returns ``(best_dist (P,), best_feat list[P] of (D,) or None, best_where list[P] of
    (batch_idx, idx_in_batch) or None)``.
    """
    P = prototype_class_identity.shape[0]
    proto_class_identity = np.argmax(prototype_class_identity, axis=1)
    proto_class_specific = np.full(P, class_specific)
    if abstain_class:
        K = num_classes - 1
        assert K >= 2, "Abstention-push must have >= 2 classes not including abstain"
        P_per_class = P // num_classes
        proto_class_specific[K * P_per_class : P] = False
    proto_dist_ = np.full(P, np.inf)
    protoL_input_: List[Optional[np.ndarray]] = [None for _ in range(P)]
    where_: List[Optional[Tuple[int, int]]] = [None for _ in range(P)]
    for push_iter, (protoL_input, proto_dist, gt) in enumerate(batches):
        for j in range(P):
            proto_dist_j = proto_dist[:, j]
            if proto_class_specific[j]:
                proto_dist_j = np.ma.masked_array(proto_dist_j, gt != proto_class_identity[j])
                if proto_dist_j.mask.all():
                    continue
            proto_dist_j_min = np.amin(proto_dist_j)
            if proto_dist_j_min <= proto_dist_[j]:
                a = int(np.argmin(proto_dist_j))
                proto_dist_[j] = proto_dist_j_min
                protoL_input_[j] = np.array(protoL_input[a, j])
                where_[j] = (push_iter, a)
    return proto_dist_, protoL_input_, where_


def xproto_push_update(protoL_input_: Sequence[Optional[np.ndarray]], prototype_shape) -> np.ndarray:
    """``prototype_vectors <- reshape(F*, prototype_shape)`` as fp32 -- push_abs_revision.py:342-346.

    Like the reference this fails when some prototype never saw its class (``None`` entries).
    """
    arr = np.array(list(protoL_input_), dtype=np.float32)
    return np.reshape(arr, tuple(prototype_shape)).astype(np.float32)


def ppnet_push_select(
    batches: Iterable[Tuple[np.ndarray, np.ndarray, np.ndarray]],
    prototype_class_identity: np.ndarray,
    num_classes: int,
    prototype_shape,
    search_batch_size: int,
    class_specific: bool = True,
    prototype_layer_stride: int = 1,
):
    """Selection loop of ``push_prototypes`` for PPNet.

    Reference: src/utils/push_ProtoPNet.py:78-92 (driver; dataset offset = push_iter * batch_size),
    :183-190 (class -> image index lists), :198-235 (per-prototype argmin over (n_c,h,w), strict ``<``,
    patch copy).

    ``batches`` yields ``(conv_output (B,D,H,W), distances (B,P,H,W), labels (B,))``.
    Returns ``(global_min_proto_dist (P,), global_min_fmap_patches (P,D,h,w), index (P,3) int64 of
    (dataset image index, h, w), -1 where never updated)``.
    """
    n_prototypes = prototype_shape[0]
    proto_h, proto_w = prototype_shape[2], prototype_shape[3]
    global_min_proto_dist = np.full(n_prototypes, np.inf)
    global_min_fmap_patches = np.zeros([n_prototypes, prototype_shape[1], proto_h, proto_w])
    index = np.full((n_prototypes, 3), -1, dtype=np.int64)
    for push_iter, (protoL_input_, proto_dist_, ys) in enumerate(batches):
        start_index_of_search_batch = push_iter * search_batch_size
        if class_specific:
            class_to_img_index_dict = {key: [] for key in range(num_classes)}
            for img_index, img_y in enumerate(np.asarray(ys)):
                class_to_img_index_dict[int(img_y)].append(img_index)
        for j in range(n_prototypes):
            if class_specific:
                target_class = int(np.argmax(prototype_class_identity[j]))
                if len(class_to_img_index_dict[target_class]) == 0:
                    continue
                proto_dist_j = proto_dist_[class_to_img_index_dict[target_class]][:, j]
            else:
                proto_dist_j = proto_dist_[:, j]
            batch_min = np.amin(proto_dist_j)
            if batch_min < global_min_proto_dist[j]:
                arg = list(np.unravel_index(np.argmin(proto_dist_j, axis=None), proto_dist_j.shape))
                if class_specific:
                    arg[0] = class_to_img_index_dict[target_class][arg[0]]
                img = arg[0]
                h0 = arg[1] * prototype_layer_stride
                w0 = arg[2] * prototype_layer_stride
                global_min_proto_dist[j] = batch_min
                global_min_fmap_patches[j] = protoL_input_[img, :, h0 : h0 + proto_h, w0 : w0 + proto_w]
                index[j] = (start_index_of_search_batch + img, arg[1], arg[2])
    return global_min_proto_dist, global_min_fmap_patches, index
