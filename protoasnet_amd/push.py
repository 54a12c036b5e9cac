"""Prototype push (projection of each prototype onto its nearest training feature), device-resident.

Drop-in for the *selection + replacement* part of the reference's two push routines
(``src/utils/push_abs_revision.py:181-348`` for XProtoNet / Video_XProtoNet,
``src/utils/push_ProtoPNet.py:14-142`` for PPNet).  Their plotting / pickling / receptive-field-box code is
out of scope; the keyword arguments that only feed it are accepted and ignored.

The reference copies every batch's features, distances, occurrence maps and input images to the host and
loops over prototypes in numpy.  Here the running (best distance, source index, source vector) per prototype
lives on the GPU and is updated by one kernel per batch (``pasn_push_*_update``); P*(D+2) words come back at
the end.  Tie rules are the reference's (later batch wins for XProtoNet, first batch wins for PPNet) and are
pinned by tests.  ``world_size > 1`` shards the loader by contiguous batches and merges with one all_gather.
"""
from __future__ import annotations

import time
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib


# ------------------------------------------------------------------------------------------------- state
class PushState:
    """Per-prototype running winners.  ``index`` is (P,) global clip indices (XProtoNet) or (P,2) = (image, s) (PPNet)."""

    def __init__(self, P: int, D: int, device, ppnet: bool = False):
        self.dist = torch.full((P,), float("inf"), dtype=torch.float32, device=device)
        self.index = torch.full((P, 2) if ppnet else (P,), -1, dtype=torch.int64, device=device)
        self.vec = torch.zeros((P, D), dtype=torch.float32, device=device)
        self.ppnet = ppnet

    def tensors(self):
        return self.dist, self.index, self.vec


def _proto_classes(model):
    ident = model.prototype_class_identity
    return torch.argmax(ident, dim=1).to(torch.int32)


def xproto_class_mask(P: int, num_classes: int, class_specific: bool, abstain_class: bool) -> torch.Tensor:
    """1 = compare only with clips of the prototype's class (push_abs_revision.py:229-237)."""
    mask = torch.full((P,), int(bool(class_specific)), dtype=torch.int32)
    if abstain_class:
        K = num_classes - 1
        assert K >= 2, "Abstention-push must have >= 2 classes not including abstain"
        mask[K * (P // num_classes):] = 0
    return mask


def merge_xproto(states: Sequence[Sequence[torch.Tensor]]):
    """Deterministic merge of per-shard winners: smallest distance; ties -> the later clip (shards hold disjoint,
    ordered batch ranges, so 'later batch wins', push_abs_revision.py:299, is 'larger global index wins')."""
    dist, index, vec = [t.clone() for t in states[0]]
    for d2, i2, v2 in states[1:]:
        take = (i2 >= 0) & ((index < 0) | (d2 < dist) | ((d2 == dist) & (i2 > index)))
        dist = torch.where(take, d2, dist)
        index = torch.where(take, i2, index)
        vec = torch.where(take[:, None], v2, vec)
    return dist, index, vec


def merge_ppnet(states: Sequence[Sequence[torch.Tensor]]):
    """Smallest distance; ties -> the earliest (image, s) (strict '<', push_ProtoPNet.py:210: first batch wins)."""
    dist, index, vec = [t.clone() for t in states[0]]
    for d2, i2, v2 in states[1:]:
        valid2, valid1 = i2[:, 0] >= 0, index[:, 0] >= 0
        earlier = (i2[:, 0] < index[:, 0]) | ((i2[:, 0] == index[:, 0]) & (i2[:, 1] < index[:, 1]))
        take = valid2 & (~valid1 | (d2 < dist) | ((d2 == dist) & earlier))
        dist = torch.where(take, d2, dist)
        index = torch.where(take[:, None], i2, index)
        vec = torch.where(take[:, None], v2, vec)
    return dist, index, vec


def _all_gather_states(state: PushState, merge):
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return state.tensors()
    world = dist.get_world_size()
    gathered = []
    for t in state.tensors():
        buf = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(buf, t.contiguous())
        gathered.append(buf)
    return merge([(gathered[0][r], gathered[1][r], gathered[2][r]) for r in range(world)])


def shard_batches(num_batches: int, rank: int, world_size: int) -> range:
    """Contiguous batch range of ``rank`` (global order stays recoverable -- SURVEY.md section 8e)."""
    per = (num_batches + world_size - 1) // world_size
    return range(min(rank * per, num_batches), min((rank + 1) * per, num_batches))


def _iter_shard(dataloader, rank: int, world_size: int):
    n = len(dataloader)
    mine = shard_batches(n, rank, world_size)
    for i, sample in enumerate(dataloader):
        if i >= mine.stop:
            break
        if i >= mine.start:
            yield i, sample


def _finish(model, state_tensors, prototype_shape, replace_prototypes: bool, log, start: float) -> Dict[str, torch.Tensor]:
    dist, index, vec = state_tensors
    if replace_prototypes:
        missing = (index.reshape(index.shape[0], -1)[:, 0] < 0).nonzero().flatten().tolist()
        if missing:
            # the reference dies here with an object-dtype reshape error (push_abs_revision.py:343-346)
            raise RuntimeError(f"push: prototypes {missing} never met a clip of their class; nothing to project onto")
        log("\tExecuting push ...")
        model.prototype_vectors.data.copy_(vec.reshape(tuple(prototype_shape)).to(model.prototype_vectors.dtype))
    log("\tpush time: \t{0}".format(time.time() - start))
    return {"proto_dist": dist, "proto_index": index, "proto_vectors": vec}


# ------------------------------------------------------------------------------------------------- XProtoNet / Video
def push_prototypes(dataloader, model, class_specific=True, abstain_class=True, preprocess_input_function=None,
                    root_dir_for_saving_prototypes=None, epoch_number=None, log=print, prototype_img_filename_prefix=None,
                    prototype_self_act_filename_prefix=None, proto_bound_boxes_filename_prefix=None, replace_prototypes=True,
                    rank: int = 0, world_size: int = 1):
    """Reference signature of ``push_abs_revision.push_prototypes``; returns the winners as device tensors."""
    model.eval()
    log(f"############## push at epoch {epoch_number} #################")
    start = time.time()
    P, D = model.num_prototypes, model.prototype_shape[1]
    device = model.prototype_vectors.device
    proto_class = _proto_classes(model).to(device)
    mask = xproto_class_mask(P, model.num_classes, class_specific, abstain_class).to(device)
    state = PushState(P, D, device)
    lib = _lib.lib()
    batch_size = getattr(dataloader, "batch_size", None)
    seen = 0
    save = root_dir_for_saving_prototypes is not None
    if save and world_size > 1:
        raise NotImplementedError("prototype artefacts (prototypes_info.pickle) are written by a single-process push; "
                                  "run the sharded sweep without root_dir_for_saving_prototypes")
    rec, names, ar = None, {}, torch.arange(P, device=device)
    for i, sample in _iter_shard(dataloader, rank, world_size):
        x = sample["cine"]
        if preprocess_input_function is not None:
            x = preprocess_input_function(x)
        labels = sample["target_AS"].to(device=device, dtype=torch.int64).contiguous()
        xdev = x.to(device)
        with torch.no_grad():
            feats, proto_dist, occ, logits = model.push_forward(xdev)
        B = int(x.shape[0])
        base = i * batch_size if batch_size else seen
        feats, proto_dist = feats.contiguous(), proto_dist.contiguous()
        _lib.check(lib.pasn_push_xproto_update(
            proto_dist.data_ptr(), feats.data_ptr(), labels.data_ptr(), proto_class.data_ptr(), mask.data_ptr(),
            state.dist.data_ptr(), state.index.data_ptr(), state.vec.data_ptr(), B, P, D, int(base), _lib.current_stream()))
        if save:
            # the records the reference keeps per prototype (push_abs_revision.py:300-307) stay ON THE DEVICE: a prototype
            # whose current winner lies in this batch takes the batch's occurrence map / logits / clip / label; no host sync
            names[int(base)] = list(sample.get("filename", [""] * B))
            here = (state.index >= base) & (state.index < base + B)
            b = (state.index - base).clamp(0, B - 1)
            new = (occ[b, ar], logits[b], xdev[b].float(), labels[b])
            if rec is None:
                rec = [torch.zeros_like(t) for t in new]
            rec = [torch.where(here.view((P,) + (1,) * (t.dim() - 1)), t, r) for t, r in zip(new, rec)]
        seen += B
    merged = _all_gather_states(state, merge_xproto) if world_size > 1 else state.tensors()
    if save and rec is not None:
        _save_xproto_artefacts(root_dir_for_saving_prototypes, epoch_number, merged, rec, names, log)
    return _finish(model, merged, model.prototype_shape, replace_prototypes, log, start)


def _save_xproto_artefacts(root: str, epoch_number, merged, rec, names, log) -> str:
    """``prototypes_info.pickle`` with the reference's keys and array shapes (push_abs_revision.py:309-325), read by
    ``explain_local`` (src/utils/local_explainability.py:36-41).  The visualisation loop that follows it in the reference
    (plots, GIFs) is not part of this package."""
    import os
    import pickle

    import numpy as np

    proto_epoch_dir = os.path.join(root, "epoch-" + str(epoch_number)) if epoch_number is not None else root
    os.makedirs(proto_epoch_dir, exist_ok=True)
    dist, index, _ = merged
    occ, logits, imgs, gts = (t.detach().cpu().numpy() for t in rec)
    idx = index.detach().cpu().numpy().astype(np.int64)
    bases = sorted(names)
    files = []
    for g in idx:
        base = max([b for b in bases if b <= g], default=None) if g >= 0 else None
        files.append(names[base][g - base] if base is not None and g - base < len(names[base]) else None)
    data = {
        "prototypes_filenames": np.array(files),
        "prototypes_src_imgs": imgs,                       # (P, 3, (To), Ho, Wo)
        "prototypes_gts": gts,                             # (P)
        "prototypes_preds": logits,                        # (P, K)
        "prototypes_occurrence_maps": occ,                 # (P, 1, (T), H, W)
        "prototypes_similarity_to_src_ROIs": 1 - dist.detach().cpu().numpy(),  # (P)
    }
    path = os.path.join(proto_epoch_dir, "prototypes_info.pickle")
    with open(path, "wb") as handle:
        pickle.dump(data, handle, protocol=pickle.HIGHEST_PROTOCOL)
    log(f"data successfully saved in {path}")
    return path


# ------------------------------------------------------------------------------------------------- PPNet
def push_prototypes_ppnet(dataloader, model, class_specific=True, preprocess_input_function=None, prototype_layer_stride=1,
                          root_dir_for_saving_prototypes=None, epoch_number=None, prototype_img_filename_prefix=None,
                          prototype_self_act_filename_prefix=None, proto_bound_boxes_filename_prefix=None,
                          save_prototype_class_identity=True, log=print, prototype_activation_function_in_numpy=None,
                          replace_prototypes=True, rank: int = 0, world_size: int = 1):
    """Reference signature of ``push_ProtoPNet.push_prototypes``; ``proto_index`` rows are (dataset image index, h*W + w)."""
    if prototype_layer_stride != 1:
        raise NotImplementedError("prototype_layer_stride != 1 is never used by the reference configs")
    model.eval()
    log("\tpush")
    start = time.time()
    P, D = model.num_prototypes, model.prototype_shape[1]
    device = model.prototype_vectors.device
    proto_class = _proto_classes(model).to(device)
    state = PushState(P, D, device, ppnet=True)
    lib = _lib.lib()
    search_batch_size = dataloader.batch_size
    for i, sample in _iter_shard(dataloader, rank, world_size):
        x = sample["cine"]
        if preprocess_input_function is not None:
            x = preprocess_input_function(x)
        labels = sample["target_AS"].to(device=device, dtype=torch.int64).contiguous()
        with torch.no_grad():
            z, (n, h, w) = model._conv_rows(x.to(device))
            _, _, dist = model._head(z, n, h * w, want_dist=True)
        _lib.check(lib.pasn_push_ppnet_update(
            dist.data_ptr(), z.data_ptr(), labels.data_ptr(), proto_class.data_ptr(), int(bool(class_specific)),
            state.dist.data_ptr(), state.index.data_ptr(), state.vec.data_ptr(), n, P, h * w, D, z.shape[-1],
            _lib.dtype_code(z.dtype), int(i * search_batch_size), _lib.current_stream()))
    merged = _all_gather_states(state, merge_ppnet) if world_size > 1 else state.tensors()
    return _finish(model, merged, model.prototype_shape, replace_prototypes, log, start)
