"""CPU: the restated push selection loops (oracle/push.py) against G4 -- the winners the REFERENCE'S OWN LOOPS chose when run
here (tests/golden/make_golden_push.py calls push_abs_revision.push_prototypes and push_ProtoPNet.push_prototypes).  The loops
are fed the reference's per-batch push_forward outputs stored in the fixture, so every assertion is exact."""
import numpy as np
import pytest
import torch

import oracle
from conftest import assert_close
from util import CFG_PUSH_PPNET, CFG_PUSH_XIMG, head_b_state, push_loader, synth_model

XCASES = ["ximg_cs_abstain", "ximg_cs", "ximg_all", "xvid_cs_abstain"]


def _xcase(g, tag):
    cs, ab, B, K = (int(v) for v in g[f"{tag}_cfg"])
    feats, dist, labels = g[f"{tag}_batch_feats"], g[f"{tag}_batch_dist"], g[f"{tag}_labels"]
    return cs, ab, B, K, [(feats[i], dist[i], labels[i]) for i in range(feats.shape[0])]


@pytest.mark.parametrize("tag", XCASES)
def test_xproto_selection_loop_equals_reference_run(golden, tag):
    g = golden("g4_push.npz")
    cs, ab, B, K, batches = _xcase(g, tag)
    P = batches[0][1].shape[1]
    ident = oracle.heads.prototype_class_identity(P, K).numpy()
    d, f, w = oracle.push.xproto_push_select(batches, ident, K, bool(cs), bool(ab))
    assert [list(x) for x in w] == g[f"{tag}_winners"].tolist()
    after = oracle.push.xproto_push_update(f, g[f"{tag}_prototypes_after"].shape)
    assert np.array_equal(after, g[f"{tag}_prototypes_after"])  # push_abs_revision.py:342-346, bit for bit
    assert np.array_equal((1 - d).astype(np.float32), g[f"{tag}_pickle_prototypes_similarity_to_src_ROIs"].astype(np.float32))
    assert not np.array_equal(after, g[f"{tag}_prototypes_before"])


def test_g4_holds_the_engineered_ties(golden):
    """'<=': the LATER of two identical batches wins in the XProtoNet push; strict '<': the FIRST wins in the PPNet push."""
    g = golden("g4_push.npz")
    w = g["ximg_all_winners"]
    assert not (w[:, 0] == 1).any() and (w[:, 0] == 4).any()  # batch 4 repeats batch 1: never batch 1
    first = g["ppnet_all_rf_boxes"][:, 0] // 4
    assert not (first == 4).any() and (first == 1).any()
    assert (g["ximg_all_winners"][:, 1] != 3)[g["ximg_all_winners"][:, 0] == 2].all()  # batch 2: image 3 repeats image 0 -> index 0


@pytest.mark.parametrize("tag", ["ppnet_cs", "ppnet_all"])
def test_ppnet_selection_loop_equals_reference_run(golden, tag):
    g = golden("g4_push.npz")
    cs, B, K = (int(v) for v in g[f"{tag}_cfg"])
    conv, dist, labels = g[f"{tag}_batch_conv"], g[f"{tag}_batch_dist"], g[f"{tag}_labels"]
    P, D = dist.shape[2], conv.shape[2]
    ident = oracle.heads.prototype_class_identity(P, K).numpy()
    batches = [(conv[i], dist[i], labels[i]) for i in range(conv.shape[0])]
    dmin, patches, index = oracle.push.ppnet_push_select(batches, ident, K, (P, D, 1, 1), B, class_specific=bool(cs))
    assert index[:, 0].tolist() == g[f"{tag}_rf_boxes"][:, 0].tolist()  # dataset image index (push_ProtoPNet.py:92,255)
    acts = g[f"{tag}_self_act"]  # the winner's activation map: its maximum is where the reference took the patch
    for j in range(P):
        h, w = np.unravel_index(np.argmax(acts[j]), acts[j].shape)
        assert (int(index[j, 1]), int(index[j, 2])) == (int(h), int(w))
    assert np.array_equal(patches.reshape(P, D, 1, 1).astype(np.float32), g[f"{tag}_prototypes_after"])
    # receptive-field boxes of the winners (columns 1-4) and their class one-hot / label columns
    rf = oracle.receptive_field
    info = [float(v) for v in g[f"{tag}_rf_info"]]
    for j in range(P):
        box = rf.rf_prototype(64, [int(index[j, 0]) % B, int(index[j, 1]), int(index[j, 2])], info)
        assert list(box[1:]) == g[f"{tag}_rf_boxes"][j, 1:5].tolist()


def test_oracle_push_forward_on_the_push_loader(golden):
    """The oracle's own model pass over the fixture's loader reproduces what the reference's push_forward returned per batch."""
    g = golden("g4_push.npz")
    sd = synth_model(CFG_PUSH_XIMG).state_dict()
    loader = push_loader("image", (3, 64, 64))
    for i, b in enumerate(loader):
        out = oracle.nets.xprotonet_forward(sd, b["cine"])
        assert_close(out["proto_dist"], g["ximg_cs_abstain_batch_dist"][i], 2e-6, 0, f"batch {i} proto_dist")
        assert_close(out["features_extracted"], g["ximg_cs_abstain_batch_feats"][i], 1e-4, 1e-4, f"batch {i} features")
        assert_close(out["logits"], g["ximg_cs_abstain_batch_logits"][i], 1e-5, 0, f"batch {i} logits")
    sdp = synth_model(CFG_PUSH_PPNET).state_dict()
    for i, b in enumerate(loader):
        out = oracle.nets.ppnet_forward(sdp, b["cine"])
        assert_close(out["conv_features"], g["ppnet_cs_batch_conv"][i], 2e-5, 2e-5, f"batch {i} conv")
        assert_close(out["distances"], g["ppnet_cs_batch_dist"][i], 1e-4, 2e-5, f"batch {i} distances")
    x = torch.stack([b["cine"] for b in push_loader("video", (48, 2, 3, 3))])
    sdv = head_b_state(48, 16, 12, 4, video=True)
    for i in range(x.shape[0]):
        out = oracle.heads.xproto_head(sdv, x[i])
        assert_close(1 - out["similarity"], g["xvid_cs_abstain_batch_dist"][i], 2e-6, 0, f"video batch {i} proto_dist")


@pytest.mark.parametrize("tag", ["ppnet_cs", "ppnet_all"])
def test_ppnet_box_files_equal_reference_run(golden, tag, tmp_path):
    """The box-file writer of the product (host code: receptive-field boxes, high-activation crop, label columns, file names) fed
    with the reference's winners and distance maps writes the arrays the reference's own run wrote (push_ProtoPNet.py:121-135).
    The bicubic upsampling is torch's on both sides here (cv2 is absent; make_golden_push.py says how), so equality is exact."""
    import types

    from protoasnet_amd import push

    g = golden("g4_push.npz")
    cs, B, K = (int(v) for v in g[f"{tag}_cfg"])
    dist = g[f"{tag}_batch_dist"]  # (nb, B, P, H, W)
    P, H, W = dist.shape[2:]
    img = g[f"{tag}_rf_boxes"][:, 0]
    acts = g[f"{tag}_self_act"]
    s = np.array([np.argmax(acts[j]) for j in range(P)])
    index = torch.from_numpy(np.stack([img, s], 1))
    dmaps = torch.from_numpy(np.stack([dist[img[j] // B, img[j] % B, j].reshape(-1) for j in range(P)]))
    labels = torch.from_numpy(np.array([g[f"{tag}_labels"][img[j] // B, img[j] % B] for j in range(P)]))
    model = types.SimpleNamespace(num_classes=K, prototype_shape=(P, 32, 1, 1), proto_layer_rf_info=[float(v) for v in g[f"{tag}_rf_info"]],
                                  prototype_activation_function="log", epsilon=1e-4)
    push._save_ppnet_artefacts(model, str(tmp_path), 3, (None, index, None), [dmaps, labels], (H, W), 64, B, True, "bb", "self_act",
                               None, lambda *a: None)
    ep = tmp_path / "epoch-3"
    assert np.array_equal(np.load(ep / "bb-receptive_field3.npy"), g[f"{tag}_rf_boxes"])
    assert np.array_equal(np.load(ep / "bb3.npy"), g[f"{tag}_bound_boxes_torch_bicubic"])
    for j in range(P):
        assert np.allclose(np.load(ep / f"self_act{j}.npy"), acts[j], atol=1e-6)
