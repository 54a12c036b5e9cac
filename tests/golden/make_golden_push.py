#!/usr/bin/env python3
"""G4: golden fixtures of the two prototype-push routines, produced by RUNNING THE REFERENCE'S OWN LOOPS on the CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_push.py

Runs only in the build container (needs /root/reference).  ``src/utils/push_abs_revision.py`` and
``src/utils/push_ProtoPNet.py`` are imported as they lie and their ``push_prototypes`` functions are *called*; nothing
of them is copied here, only their numerical results are stored (``g4_push.npz``).

What this image lacks and how the import still succeeds (ordinary ModuleNotFoundErrors, no permission was denied):

* ``cv2``, ``moviepy``, ``imageio``, ``torchvision`` are absent: empty placeholder modules are registered.  The XProtoNet
  routine touches cv2 / moviepy only inside ``prototype_plot`` (visualisation), which is replaced by a no-op.  The PPNet
  routine calls ``cv2.resize(act, (S, S), interpolation=cv2.INTER_CUBIC)`` inside its selection loop to derive the
  high-activation bounding box: the placeholder implements it with torch's bicubic interpolation (same a = -0.75 kernel,
  half-pixel centres, edge replication as OpenCV).  Winners' indices, patches, distances and receptive-field boxes do not
  depend on it; the ``proto_bound_boxes`` rows do and are stored under ``*_bound_boxes_torch_bicubic`` for that reason.
* There is no GPU: ``Tensor.cuda`` / ``Module.cuda`` are patched to identity (``push_abs_revision.py:268,346``).

The per-batch ``push_forward`` outputs of the reference model are stored too, so the restated selection loops
(oracle/push.py) and the device kernels (pasn_push_*_update) can be fed EXACTLY what the reference loops saw: any
disagreement in a winner is then a tie-rule / masking error, never floating-point noise.  Engineered cases: a whole
batch repeated later in the loader (exact tie across batches: '<=' keeps the later one in push_abs_revision.py:299,
'<' keeps the first in push_ProtoPNet.py:210), an image repeated inside a batch (np.argmin keeps the first),
a batch in which some classes are absent, abstain prototypes that ignore the labels.
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PASN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)
    return sys.modules[name]


def _bicubic_resize(img, dsize, interpolation=None):
    t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))[None, None]
    out = torch.nn.functional.interpolate(t, size=(int(dsize[1]), int(dsize[0])), mode="bicubic", align_corners=False)
    return out[0, 0].numpy()


_tvm = _placeholder("torchvision.models")
_placeholder("torchvision", models=_tvm)
_placeholder("cv2", resize=_bicubic_resize, INTER_CUBIC=2)
_placeholder("imageio")
_isc = _placeholder("moviepy.video.io.ImageSequenceClip", ImageSequenceClip=object)
_vio = _placeholder("moviepy.video.io", ImageSequenceClip=_isc)
_vid = _placeholder("moviepy.video", io=_vio)
_placeholder("moviepy.editor", ImageSequenceClip=object)
_placeholder("moviepy", video=_vid)
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

from src.models.ProtoPNet import construct_PPNet  # noqa: E402  (reference)
from src.models.XProtoNet import construct_XProtoNet  # noqa: E402  (reference)
from src.models.Video_XProtoNet import Video_XProtoNet  # noqa: E402  (reference)
import src.utils.push_abs_revision as ref_push_x  # noqa: E402  (reference)
import src.utils.push_ProtoPNet as ref_push_p  # noqa: E402  (reference)

from protoasnet_amd import synth  # noqa: E402

ref_push_x.prototype_plot = lambda *a, **k: None  # visualisation only (cv2 / matplotlib / moviepy)
torch.manual_seed(0)
torch.set_num_threads(8)


class Loader:
    """What the reference loops need of a DataLoader: len, iteration over dict samples, ``batch_size``."""

    def __init__(self, batches, batch_size):
        self.batches, self.batch_size = batches, batch_size

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def push_recipe(kind):
    """(seed offsets, labels) of the loader -- the tests rebuild the very same loader from this table (tests/util.py)."""
    if kind == "image":
        # batch 4 repeats batch 1 (tie across batches); image 3 of batch 2 repeats image 0 (tie inside a batch);
        # batch 3 holds class 1 only (other classes absent in that batch)
        seeds = [[10, 11, 12, 13], [20, 21, 22, 23], [30, 31, 32, 30], [40, 41, 42, 43], [20, 21, 22, 23], [50, 51, 52, 53]]
        labels = [[0, 1, 2, 0], [1, 2, 0, 1], [2, 2, 1, 2], [1, 1, 1, 1], [1, 2, 0, 1], [0, 0, 2, 1]]
    else:
        seeds = [[110, 111, 112], [120, 121, 122], [130, 131, 130], [120, 121, 122], [140, 141, 142]]
        labels = [[0, 1, 2], [2, 0, 1], [1, 1, 1], [2, 0, 1], [0, 2, 2]]
    return seeds, labels


def make_batches(kind, shape):
    seeds, labels = push_recipe(kind)
    out = []
    for bi, (ss, ls) in enumerate(zip(seeds, labels)):
        if kind == "video":  # the stand-in trunk passes pre-made post-ReLU features through (as in G3)
            x = torch.stack([torch.from_numpy(np.maximum(np.random.default_rng(s).standard_normal(shape).astype(np.float32), 0.0)) for s in ss])
        else:
            x = torch.cat([synth.echo_clips((1,) + shape, seed=s) for s in ss])
        out.append({"cine": x, "target_AS": torch.tensor(ls, dtype=torch.int64), "filename": [f"b{bi}_{a}" for a in range(len(ss))]})
    return out


class resnet2p1d_18(torch.nn.Module):  # noqa: N801 -- the reference sniffs this class name (ProtoPNet.py:152-156)
    def __init__(self, channels):
        super().__init__()
        self.probe = torch.nn.Conv3d(channels, channels, kernel_size=1, bias=False)

    def forward(self, x):
        return x


def npy(t):
    return t.detach().cpu().numpy()


def run_xproto(tag, model, batches, batch_size, class_specific, abstain, out):
    model.eval()
    per_batch = []
    with torch.no_grad():
        for b in batches:
            f, d, occ, logits = model.push_forward(b["cine"])
            per_batch.append((npy(f), npy(d), npy(occ), npy(logits)))
    before = npy(model.prototype_vectors).copy()
    with tempfile.TemporaryDirectory() as tmp:
        ref_push_x.push_prototypes(Loader(batches, batch_size), model, class_specific=class_specific, abstain_class=abstain,
                                   root_dir_for_saving_prototypes=tmp, epoch_number=7, log=lambda *a: None, replace_prototypes=True)
        with open(os.path.join(tmp, "epoch-7", "prototypes_info.pickle"), "rb") as fh:
            info = pickle.load(fh)
    winners = np.array([[int(s.split("_")[0][1:]), int(s.split("_")[1])] for s in info["prototypes_filenames"]], dtype=np.int64)
    out[f"{tag}_cfg"] = np.array([int(class_specific), int(abstain), batch_size, model.num_classes], dtype=np.int64)
    out[f"{tag}_batch_feats"] = np.stack([p[0] for p in per_batch])      # (nb, B, P, D)  reference push_forward outputs
    out[f"{tag}_batch_dist"] = np.stack([p[1] for p in per_batch])       # (nb, B, P)
    out[f"{tag}_batch_logits"] = np.stack([p[3] for p in per_batch])     # (nb, B, K)
    out[f"{tag}_labels"] = np.stack([npy(b["target_AS"]) for b in batches])
    out[f"{tag}_winners"] = winners                                        # (P, 2) = (batch, index in batch) chosen by the reference loop
    out[f"{tag}_prototypes_before"] = before
    out[f"{tag}_prototypes_after"] = npy(model.prototype_vectors)         # after push_abs_revision.py:342-346
    for k in ("prototypes_gts", "prototypes_preds", "prototypes_occurrence_maps", "prototypes_similarity_to_src_ROIs"):
        out[f"{tag}_pickle_{k}"] = np.asarray(info[k])
    out[f"{tag}_pickle_src_imgs_sum"] = np.asarray(info["prototypes_src_imgs"], dtype=np.float64).reshape(len(winners), -1).sum(1)
    out[f"{tag}_pickle_keys"] = np.array(sorted(info.keys()))
    d = np.stack([p[1] for p in per_batch])
    gaps = np.sort(d.reshape(-1, d.shape[-1]), axis=0)
    print(tag, "winners", winners.tolist(), "| min top-2 gap of distinct values per prototype",
          float(min(np.diff(np.unique(gaps[:, j]))[0] for j in range(d.shape[-1]))))


def run_ppnet(tag, model, batches, batch_size, class_specific, out):
    model.eval()
    per_batch = []
    with torch.no_grad():
        for b in batches:
            conv, dist = model.push_forward(b["cine"])
            per_batch.append((npy(conv), npy(dist)))
    before = npy(model.prototype_vectors).copy()
    P = model.num_prototypes
    with tempfile.TemporaryDirectory() as tmp:
        ref_push_p.push_prototypes(Loader(batches, batch_size), model, class_specific=class_specific, root_dir_for_saving_prototypes=tmp,
                                   epoch_number=3, prototype_img_filename_prefix=None, prototype_self_act_filename_prefix="self_act",
                                   proto_bound_boxes_filename_prefix="bb", save_prototype_class_identity=True, log=lambda *a: None)
        ep = os.path.join(tmp, "epoch-3")
        rf = np.load(os.path.join(ep, "bb-receptive_field3.npy"))
        bb = np.load(os.path.join(ep, "bb3.npy"))
        acts = np.stack([np.load(os.path.join(ep, f"self_act{j}.npy")) for j in range(P)])
    out[f"{tag}_cfg"] = np.array([int(class_specific), batch_size, model.num_classes], dtype=np.int64)
    out[f"{tag}_batch_conv"] = np.stack([p[0] for p in per_batch])        # (nb, B, D, H, W)
    out[f"{tag}_batch_dist"] = np.stack([p[1] for p in per_batch])        # (nb, B, P, H, W)
    out[f"{tag}_labels"] = np.stack([npy(b["target_AS"]) for b in batches])
    out[f"{tag}_prototypes_before"] = before
    out[f"{tag}_prototypes_after"] = npy(model.prototype_vectors)          # after push_ProtoPNet.py:137-140
    out[f"{tag}_rf_boxes"] = rf                                             # bb-receptive_field<epoch>.npy (push_ProtoPNet.py:121-128)
    out[f"{tag}_bound_boxes_torch_bicubic"] = bb                            # bb<epoch>.npy (:129-135); box from the placeholder resize
    out[f"{tag}_self_act"] = acts                                           # <prefix><j>.npy: activation map of the winning image
    out[f"{tag}_rf_info"] = np.array(model.proto_layer_rf_info, dtype=np.float64)
    print(tag, "rf_boxes[:,0] (dataset image index of each winner)", rf[:, 0].tolist())


def main():
    out = {}
    # ---- XProtoNet (image), ResNet-18 trunk, 64x64 images -> 2x2 feature map; abstain + class-specific (the shipped setting)
    img_batches = make_batches("image", (3, 64, 64))
    for tag, cs, ab in (("ximg_cs_abstain", True, True), ("ximg_cs", True, False), ("ximg_all", False, False)):
        m = construct_XProtoNet("resnet18", pretrained=False, img_size=64, prototype_shape=(12, 32, 1, 1), num_classes=4 if ab else 3,
                                add_on_layers_type="regular")
        synth.load_synth(m)
        run_xproto(tag, m, img_batches, 4, cs, ab, out)
    # ---- Video_XProtoNet head over pre-made trunk features (stand-in trunk as in G3), abstain setting of Ours_ProtoASNet_Video.yml
    vid_batches = make_batches("video", (48, 2, 3, 3))
    m = Video_XProtoNet(cnn_backbone=resnet2p1d_18(48), img_size=112, prototype_shape=(12, 16, 1, 1, 1), proto_layer_rf_info=None,
                        num_classes=4, init_weights=True)
    synth.load_synth(m)
    run_xproto("xvid_cs_abstain", m, vid_batches, 3, True, True, out)
    # ---- PPNet, ResNet-18 trunk, 64x64 -> 2x2 map
    for tag, cs in (("ppnet_cs", True), ("ppnet_all", False)):
        m = construct_PPNet("resnet18", pretrained=False, img_size=64, prototype_shape=(6, 32, 1, 1), num_classes=3,
                            prototype_activation_function="log", add_on_layers_type="regular")
        synth.load_synth(m)
        run_ppnet(tag, m, img_batches, 4, cs, out)
    path = os.path.join(HERE, "g4_push.npz")
    np.savez_compressed(path, **out)
    print(f"g4_push.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


if __name__ == "__main__":
    main()
