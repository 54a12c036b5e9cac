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


def _require_group(world_size: int) -> None:
    """A sharded push (``world_size > 1``) needs the process group it will merge over; pushing from one shard only, silently,
    would project the prototypes onto a fraction of the data."""
    import torch.distributed as dist

    if world_size <= 1:
        return
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError(f"push with world_size={world_size} needs torch.distributed to be initialised (one process per GPU)")
    if dist.get_world_size() != world_size:
        raise RuntimeError(f"push: world_size={world_size} but the process group has {dist.get_world_size()} ranks")


def _all_gather_states(state: PushState, merge):
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("_all_gather_states needs an initialised process group")
    world = dist.get_world_size()
    if world == 1:
        return state.tensors()
    gathered = []
    for t in state.tensors():
        t = _for_collective(t.contiguous())
        buf = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(buf, t)
        gathered.append([b.to(state.dist.device) for b in buf])
    return merge([(gathered[0][r], gathered[1][r], gathered[2][r]) for r in range(world)])


def _for_collective(t: torch.Tensor) -> torch.Tensor:
    """RCCL takes device tensors; the gloo rehearsal backend (several ranks on one card, CPU tests) takes host tensors."""
    import torch.distributed as dist

    return t.cpu() if (t.is_cuda and dist.get_backend() == "gloo") else t


def shard_batches(num_batches: int, rank: int, world_size: int) -> range:
    """Contiguous batch range of ``rank`` (global order stays recoverable -- SURVEY.md section 8e)."""
    per = (num_batches + world_size - 1) // world_size
    return range(min(rank * per, num_batches), min((rank + 1) * per, num_batches))


def _labels_to_device(labels, device, pinned: list) -> torch.Tensor:
    """The batch's labels as int64 on the device WITHOUT stalling the host: a pageable host tensor's ``.to(device)`` returns only
    when the copy has run, i.e. after every launch already queued -- per batch that drained the queue and left the GPU idle while the
    next batch's ~80 launches were enqueued (push sweep over 10 000 clips: 8.8 k clips/s against 10.5 k for the bare forward).  Labels go
    through a pinned staging tensor and a non-blocking copy; ``pinned`` keeps the last few staging tensors alive until their copies ran."""
    labels = torch.as_tensor(labels)
    if labels.device.type != "cpu" or torch.device(device).type == "cpu":
        return labels.to(device=device, dtype=torch.int64).contiguous()
    stage = labels.to(torch.int64).contiguous().pin_memory()
    pinned.append(stage)
    del pinned[:-8]
    return stage.to(device, non_blocking=True)


def _iter_shard(dataloader, rank: int, world_size: int):
    """Yields ``(batch index, global index of the batch's first clip, sample)`` for this rank's contiguous batch range.

    A ``torch.utils.data.DataLoader`` is re-instantiated over the rank's slice of its ``batch_sampler`` (same dataset,
    collate function and workers), so foreign batches are never loaded and the clip offsets come from a prefix sum of the
    sampler's batch lengths.  Any other iterable of samples is walked from the start: batches before the range are consumed
    to count their clips (the only way to know the offset of a ragged loader), batches after it are not touched."""
    n = len(dataloader)
    mine = shard_batches(n, rank, world_size)
    if isinstance(dataloader, torch.utils.data.DataLoader) and dataloader.batch_sampler is not None:
        inner = getattr(dataloader.batch_sampler, "sampler", None)
        if world_size > 1 and inner is not None and not isinstance(inner, torch.utils.data.SequentialSampler):
            # every rank iterates the batch sampler on its own: a random sampler would hand each rank a different permutation, the
            # contiguous shards would overlap / miss clips and the global indices would not name the same clips across ranks
            raise ValueError("sharded push needs a push loader with a SequentialSampler (shuffle=False, as the reference's push loader is); "
                             f"got {type(inner).__name__}")
        index_lists = [list(b) for b in dataloader.batch_sampler]
        offsets = np.concatenate([[0], np.cumsum([len(b) for b in index_lists])])
        sub = torch.utils.data.DataLoader(dataloader.dataset, batch_sampler=index_lists[mine.start:mine.stop],
                                          num_workers=dataloader.num_workers, collate_fn=dataloader.collate_fn,
                                          pin_memory=dataloader.pin_memory)
        for k, sample in enumerate(sub):
            yield mine.start + k, int(offsets[mine.start + k]), sample
        return
    batch_size = getattr(dataloader, "batch_size", None)
    seen = 0
    for i, sample in enumerate(dataloader):
        if i >= mine.stop:
            break
        if i >= mine.start:
            yield i, (i * batch_size if batch_size else seen), sample
        seen += int(sample["cine"].shape[0])


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
    _require_group(world_size)
    model.eval()
    log(f"############## push at epoch {epoch_number} #################")
    start = time.time()
    P, D = model.num_prototypes, model.prototype_shape[1]
    device = model.prototype_vectors.device
    proto_class = _proto_classes(model).to(device)
    mask = xproto_class_mask(P, model.num_classes, class_specific, abstain_class).to(device)
    state = PushState(P, D, device)
    lib = _lib.lib()
    save = root_dir_for_saving_prototypes is not None
    rec, names, ar = None, {}, torch.arange(P, device=device)
    pinned = []
    for i, base, sample in _iter_shard(dataloader, rank, world_size):
        x = sample["cine"]
        if preprocess_input_function is not None:
            x = preprocess_input_function(x)
        labels = _labels_to_device(sample["target_AS"], device, pinned)
        xdev = x.to(device)
        with torch.no_grad():
            feats, proto_dist, occ, logits = model.push_forward(xdev)
        B = int(x.shape[0])
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
    local_index = state.index.clone()
    merged = _all_gather_states(state, merge_xproto) if world_size > 1 else state.tensors()
    if save:
        if world_size > 1:
            rec, names = _reduce_records(rec, names, local_index, merged[1], device)
        if rank == 0 and rec is not None:
            _save_xproto_artefacts(root_dir_for_saving_prototypes, epoch_number, merged, rec, names, log)
    return _finish(model, merged, model.prototype_shape, replace_prototypes, log, start)


def _reduce_records(rec, names, local_index, merged_index, device):
    """Sharded push: the records of a prototype live on the rank whose shard holds its final winner.  Every rank zeroes the
    records it does not own, ONE sum-reduce per record tensor brings them to rank 0 (exact: all other addends are zeros), the
    file names travel as objects.  Shapes are agreed on first (a rank with an empty shard has no records yet)."""
    import torch.distributed as dist

    shapes = [None] * dist.get_world_size()
    dist.all_gather_object(shapes, None if rec is None else [(tuple(t.shape), str(t.dtype)) for t in rec])
    proto = next((sh for sh in shapes if sh is not None), None)
    if proto is None:
        return None, names
    if rec is None:
        rec = [torch.zeros(sh, dtype=getattr(torch, dt.split(".")[1]), device=device) for sh, dt in proto]
    own = (local_index >= 0) & (local_index == merged_index)
    out = []
    for t in rec:
        t = torch.where(own.view((-1,) + (1,) * (t.dim() - 1)), t, torch.zeros_like(t))
        t = _for_collective(t.contiguous())
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        out.append(t.to(device))
    all_names = [None] * dist.get_world_size()
    dist.all_gather_object(all_names, names)
    merged_names = {}
    for d in all_names:
        merged_names.update(d)
    return out, merged_names


def _save_xproto_artefacts(root: str, epoch_number, merged, rec, names, log) -> str:
    """``prototypes_info.pickle`` with the reference's keys and array shapes (push_abs_revision.py:309-325), read by
    ``explain_local`` (src/utils/local_explainability.py:36-41).  The visualisation loop that follows it in the reference
    (plots, GIFs) is not part of this package."""
    import os
    import pickle

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
def find_high_activation_crop(activation_map: np.ndarray, percentile: float = 95):
    """(y0, y1, x0, x1) of the rows / columns holding a value at or above the percentile (src/utils/utils.py:259-280)."""
    mask = activation_map >= np.percentile(activation_map, percentile)
    rows, cols = np.nonzero(mask.any(axis=1))[0], np.nonzero(mask.any(axis=0))[0]
    if rows.size == 0:
        return 0, 1, 0, 1
    return int(rows[0]), int(rows[-1]) + 1, int(cols[0]), int(cols[-1]) + 1


def push_prototypes_ppnet(dataloader, model, class_specific=True, preprocess_input_function=None, prototype_layer_stride=1,
                          root_dir_for_saving_prototypes=None, epoch_number=None, prototype_img_filename_prefix=None,
                          prototype_self_act_filename_prefix=None, proto_bound_boxes_filename_prefix=None,
                          save_prototype_class_identity=True, log=print, prototype_activation_function_in_numpy=None,
                          replace_prototypes=True, rank: int = 0, world_size: int = 1):
    """Reference signature of ``push_ProtoPNet.push_prototypes``; ``proto_index`` rows are (dataset image index, h*W + w).

    With a saving directory, the box files of the reference are written (push_ProtoPNet.py:121-135):
    ``<prefix>-receptive_field<epoch>.npy`` and ``<prefix><epoch>.npy`` (rows ``[dataset image index, y0, y1, x0, x1, label x K]``,
    -1 where a prototype was never updated) and, with ``prototype_self_act_filename_prefix``, the winners' activation maps
    ``<prefix><j>.npy``.  The winner's distance map and label stay on the device during the sweep (one ``torch.where`` per
    batch, no per-batch copy to the host); the boxes are derived once at the end.  The PNG renderings are out of scope."""
    if prototype_layer_stride != 1:
        raise NotImplementedError("prototype_layer_stride != 1 is never used by the reference configs")
    _require_group(world_size)
    model.eval()
    log("\tpush")
    start = time.time()
    P, D = model.num_prototypes, model.prototype_shape[1]
    device = model.prototype_vectors.device
    proto_class = _proto_classes(model).to(device)
    state = PushState(P, D, device, ppnet=True)
    lib = _lib.lib()
    search_batch_size = dataloader.batch_size
    save = root_dir_for_saving_prototypes is not None
    rec, hw, img_side = None, None, None
    pinned = []
    for i, _, sample in _iter_shard(dataloader, rank, world_size):
        x = sample["cine"]
        if preprocess_input_function is not None:
            x = preprocess_input_function(x)
        labels = _labels_to_device(sample["target_AS"], device, pinned)
        with torch.no_grad():
            z, (n, h, w) = model._conv_rows(x.to(device))
            _, _, dist = model._head(z, n, h * w, want_dist=True)
        base = int(i * search_batch_size)  # push_ProtoPNet.py:92
        _lib.check(lib.pasn_push_ppnet_update(
            dist.data_ptr(), z.data_ptr(), labels.data_ptr(), proto_class.data_ptr(), int(bool(class_specific)),
            state.dist.data_ptr(), state.index.data_ptr(), state.vec.data_ptr(), n, P, h * w, D, z.shape[-1],
            _lib.dtype_code(z.dtype), base, _lib.current_stream()))
        if save:
            hw, img_side = (h, w), int(x.shape[2])
            img = state.index[:, 0]
            here = (img >= base) & (img < base + n)
            b = (img - base).clamp(0, n - 1)
            new = (dist[b, torch.arange(P, device=device)], labels[b])  # (P, h*w) distance map of the winner image, (P,) its label
            if rec is None:
                rec = [torch.zeros_like(t) for t in new]
            rec = [torch.where(here.view((P,) + (1,) * (t.dim() - 1)), t, r) for t, r in zip(new, rec)]
    local_index = state.index.clone()
    merged = _all_gather_states(state, merge_ppnet) if world_size > 1 else state.tensors()
    if save:
        if world_size > 1:
            import torch.distributed as dist_

            meta = [None] * world_size
            dist_.all_gather_object(meta, (hw, img_side))
            hw, img_side = next((m for m in meta if m[0] is not None), (None, None))
            own_index, merged_img = local_index[:, 0], merged[1][:, 0]
            rec, _ = _reduce_records(rec, {}, torch.where((local_index == merged[1]).all(1), own_index, torch.full_like(own_index, -1)),
                                     merged_img, device)
        if rank == 0 and rec is not None:
            _save_ppnet_artefacts(model, root_dir_for_saving_prototypes, epoch_number, merged, rec, hw, img_side, search_batch_size,
                                  save_prototype_class_identity, proto_bound_boxes_filename_prefix,
                                  prototype_self_act_filename_prefix, prototype_activation_function_in_numpy, log)
    return _finish(model, merged, model.prototype_shape, replace_prototypes, log, start)


def _save_ppnet_artefacts(model, root, epoch_number, merged, rec, hw, img_side, batch_size, with_class, box_prefix, act_prefix,
                          act_fn_numpy, log) -> None:
    import os

    from .receptive_field import compute_rf_prototype

    proto_epoch_dir = os.path.join(root, "epoch-" + str(epoch_number)) if epoch_number is not None else root
    os.makedirs(proto_epoch_dir, exist_ok=True)
    _, index, _ = merged
    index = index.detach().cpu().numpy().astype(np.int64)
    dmaps = rec[0].detach().float().cpu().numpy().reshape(index.shape[0], hw[0], hw[1])
    labels = rec[1].detach().cpu().numpy()
    P, K = index.shape[0], model.num_classes
    cols = 5 + K if with_class else 5
    rf_boxes = np.full((P, cols), -1)
    bound_boxes = np.full((P, cols), -1)
    D = model.prototype_shape[1] * model.prototype_shape[2] * model.prototype_shape[3]
    for j in range(P):
        if index[j, 0] < 0:
            continue
        h, w = divmod(int(index[j, 1]), hw[1])
        box = compute_rf_prototype(img_side, [int(index[j, 0]) % batch_size, h, w], model.proto_layer_rf_info)
        rf_boxes[j, 0] = index[j, 0]
        rf_boxes[j, 1:5] = box[1:]
        d = dmaps[j]
        if model.prototype_activation_function == "log":
            act = np.log((d + 1) / (d + model.epsilon))
        elif model.prototype_activation_function == "linear":
            act = D - d
        else:
            act = act_fn_numpy(d)
        # the reference upsamples with cv2.resize(INTER_CUBIC); torch's bicubic uses the same a = -0.75 kernel, half-pixel
        # centres and edge replication (a box edge can differ by a pixel where the percentile threshold is met marginally)
        up = torch.nn.functional.interpolate(torch.from_numpy(np.ascontiguousarray(act, dtype=np.float32))[None, None],
                                             size=(img_side, img_side), mode="bicubic", align_corners=False)[0, 0].numpy()
        bound_boxes[j, 0] = index[j, 0]
        bound_boxes[j, 1:5] = find_high_activation_crop(up)
        if with_class:
            rf_boxes[j, 5:] = labels[j]
            bound_boxes[j, 5:] = labels[j]
        if act_prefix is not None:
            np.save(os.path.join(proto_epoch_dir, act_prefix + str(j) + ".npy"), act)
    if box_prefix is not None:
        np.save(os.path.join(proto_epoch_dir, box_prefix + "-receptive_field" + str(epoch_number) + ".npy"), rf_boxes)
        np.save(os.path.join(proto_epoch_dir, box_prefix + str(epoch_number) + ".npy"), bound_boxes)
        log(f"\tprototype boxes saved in {proto_epoch_dir}")
