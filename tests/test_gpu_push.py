"""GPU: the device-resident push sweeps vs the restated reference selection loops -- indices must be bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from protoasnet_amd import synth
from util import CFG_PPNET_BOTTLENECK, CFG_VIDEO_X3D, synth_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _xproto_sweep(batches, ident, num_classes, class_specific, abstain, P, D, batch_size):
    from protoasnet_amd import _lib
    from protoasnet_amd.push import PushState, xproto_class_mask

    state = PushState(P, D, DEV)
    cls = torch.argmax(torch.from_numpy(ident), dim=1).to(torch.int32).to(DEV)
    mask = xproto_class_mask(P, num_classes, class_specific, abstain).to(DEV)
    for i, (feat, dist, gt) in enumerate(batches):
        f, d, g = torch.from_numpy(feat).to(DEV), torch.from_numpy(dist).to(DEV), torch.from_numpy(gt).to(DEV)
        _lib.check(_lib.lib().pasn_push_xproto_update(d.data_ptr(), f.data_ptr(), g.data_ptr(), cls.data_ptr(), mask.data_ptr(),
                                                      state.dist.data_ptr(), state.index.data_ptr(), state.vec.data_ptr(),
                                                      feat.shape[0], P, D, i * batch_size, 0))
    torch.cuda.synchronize()
    return state.dist.cpu().numpy(), state.index.cpu().numpy(), state.vec.cpu().numpy()


@pytest.mark.parametrize("class_specific,abstain", [(True, True), (True, False), (False, False)])
def test_xproto_push_kernel_bit_exact(class_specific, abstain):
    rng = np.random.default_rng(5)
    P, D, K, B, nb = 40, 32, 4, 7, 9
    ident = oracle.heads.prototype_class_identity(P, K).numpy()
    batches = []
    for b in range(nb):
        dist = rng.random((B, P)).astype(np.float32)
        dist = np.round(dist * 8) / 8  # many exact ties inside and across batches
        gt = rng.integers(0, 3, size=B).astype(np.int64)
        if b == 2:
            gt[:] = 1  # a batch where some classes are absent (mask.all() -> continue)
        batches.append((rng.standard_normal((B, P, D)).astype(np.float32), dist, gt))
    ref_d, ref_f, ref_w = oracle.push.xproto_push_select(batches, ident, K, class_specific, abstain)
    d, idx, vec = _xproto_sweep(batches, ident, K, class_specific, abstain, P, D, B)
    assert any(w is None for w in ref_w) == (class_specific and not abstain)  # labels are 0..2: class 3 is never seen
    for j in range(P):
        if ref_w[j] is None:  # the prototype's class never appeared: the state must be untouched
            assert idx[j] == -1 and np.isinf(d[j])
            continue
        assert idx[j] == ref_w[j][0] * B + ref_w[j][1], f"prototype {j}: {idx[j]} vs {ref_w[j]}"
        assert d[j] == np.float32(ref_d[j])
        assert np.array_equal(vec[j], ref_f[j])


def test_xproto_push_never_seen_class_stays_empty():
    P, D, K = 8, 4, 4
    ident = oracle.heads.prototype_class_identity(P, K).numpy()
    feat = np.ones((3, P, D), np.float32)
    dist = np.full((3, P), 0.25, np.float32)
    gt = np.zeros(3, np.int64)  # only class 0 present
    d, idx, _ = _xproto_sweep([(feat, dist, gt)], ident, K, True, True, P, D, 3)
    assert idx.tolist() == [0, 0, -1, -1, -1, -1, 0, 0]  # classes 1,2 unseen; abstain prototypes (last P/K) unmasked
    assert np.isinf(d[2:6]).all()


def _ppnet_sweep(batches, ident, num_classes, class_specific, P, D, S, batch_size):
    from protoasnet_amd import _lib
    from protoasnet_amd.push import PushState

    state = PushState(P, D, DEV, ppnet=True)
    cls = torch.argmax(torch.from_numpy(ident), dim=1).to(torch.int32).to(DEV)
    for i, (conv, dist, y) in enumerate(batches):
        B = conv.shape[0]
        z = torch.from_numpy(conv).reshape(B, D, S).transpose(1, 2).contiguous().to(DEV)  # [B][S][D], D % 8 == 0
        dd = torch.from_numpy(dist).reshape(B, P, S).contiguous().to(DEV)
        yy = torch.from_numpy(y).to(DEV)
        _lib.check(_lib.lib().pasn_push_ppnet_update(dd.data_ptr(), z.data_ptr(), yy.data_ptr(), cls.data_ptr(), int(class_specific),
                                                     state.dist.data_ptr(), state.index.data_ptr(), state.vec.data_ptr(), B, P, S, D,
                                                     D, 0, i * batch_size, 0))
    torch.cuda.synchronize()
    return state.dist.cpu().numpy(), state.index.cpu().numpy(), state.vec.cpu().numpy()


@pytest.mark.parametrize("class_specific", [True, False])
def test_ppnet_push_kernel_bit_exact(class_specific):
    rng = np.random.default_rng(6)
    P, D, K, B, H, W, nb = 30, 16, 3, 5, 4, 3, 6
    ident = oracle.heads.prototype_class_identity(P, K).numpy()
    batches = []
    for b in range(nb):
        dist = np.round(rng.random((B, P, H, W)).astype(np.float32) * 16) / 16
        y = rng.integers(0, K, size=B).astype(np.int64)
        if b == 1:
            y[:] = 2
        batches.append((rng.standard_normal((B, D, H, W)).astype(np.float32), dist, y))
    ref_d, ref_p, ref_i = oracle.push.ppnet_push_select(batches, ident, K, (P, D, 1, 1), B, class_specific)
    d, idx, vec = _ppnet_sweep(batches, ident, K, class_specific, P, D, H * W, B)
    assert np.array_equal(idx[:, 0], ref_i[:, 0])
    assert np.array_equal(idx[:, 1], ref_i[:, 1] * W + ref_i[:, 2])
    assert np.array_equal(d, ref_d.astype(np.float32))
    assert np.array_equal(vec, ref_p[:, :, 0, 0].astype(np.float32))


class _Loader(list):
    batch_size = 4


def _loader(shape, nb, seed):
    out = _Loader()
    for b in range(nb):
        out.append({"cine": synth.echo_clips((4,) + shape, seed=seed + b), "target_AS": (torch.arange(4) + b) % 3,
                    "filename": [f"c{b}_{i}" for i in range(4)]})
    return out


def test_push_prototypes_video_end_to_end():
    """push_prototypes over a loader (reference signature) == restated reference loop fed with the same push_forward outputs."""
    from protoasnet_amd.push import push_prototypes

    m = synth_model(CFG_VIDEO_X3D).to(DEV).eval()
    loader = _loader((3, 4, 64, 64), nb=5, seed=40)
    recorded = []
    with torch.no_grad():
        for s in loader:
            f, d, _, _ = m.push_forward(s["cine"].to(DEV))
            recorded.append((f.cpu().numpy(), d.cpu().numpy(), s["target_AS"].numpy()))
    ref_d, ref_f, ref_w = oracle.push.xproto_push_select(recorded, m.prototype_class_identity.numpy(), m.num_classes, True, False)
    out = push_prototypes(loader, m, class_specific=True, abstain_class=False, replace_prototypes=True, log=lambda *_: None)
    idx = out["proto_index"].cpu().numpy()
    assert [int(i) for i in idx] == [w[0] * 4 + w[1] for w in ref_w]
    assert np.array_equal(out["proto_dist"].cpu().numpy(), ref_d.astype(np.float32))
    want = oracle.push.xproto_push_update(ref_f, m.prototype_shape)
    assert np.array_equal(m.prototype_vectors.detach().cpu().numpy(), want)
    # idempotence: a second push over the same data keeps the same winners (later-batch tie rule included)
    with torch.no_grad():
        sim = m.push_forward(loader[ref_w[0][0]]["cine"].to(DEV))[1]
    assert float(sim[ref_w[0][1], 0]) < 1e-5  # the projected prototype now sits on its source feature


def test_push_prototypes_writes_reference_pickle(tmp_path):
    """root_dir_for_saving_prototypes: prototypes_info.pickle with the reference's keys (push_abs_revision.py:309-325); the
    records are those of the winning clips, computed with the prototypes as they were BEFORE the projection."""
    import pickle

    from protoasnet_amd.push import push_prototypes

    m = synth_model(CFG_VIDEO_X3D).to(DEV).eval()
    loader = _loader((3, 4, 64, 64), nb=4, seed=70)
    rec = []
    with torch.no_grad():
        for s in loader:
            f, d, o, l = m.push_forward(s["cine"].to(DEV))
            rec.append((d.cpu(), o.cpu(), l.cpu()))
    out = push_prototypes(loader, m, class_specific=True, abstain_class=False, replace_prototypes=True, log=lambda *_: None,
                          root_dir_for_saving_prototypes=str(tmp_path), epoch_number=3)
    with open(tmp_path / "epoch-3" / "prototypes_info.pickle", "rb") as fh:
        info = pickle.load(fh)
    assert sorted(info) == sorted(["prototypes_filenames", "prototypes_src_imgs", "prototypes_gts", "prototypes_preds",
                                   "prototypes_occurrence_maps", "prototypes_similarity_to_src_ROIs"])
    P = m.num_prototypes
    assert info["prototypes_src_imgs"].shape == (P, 3, 4, 64, 64) and info["prototypes_occurrence_maps"].shape == (P, 1, 4, 2, 2)
    assert info["prototypes_preds"].shape == (P, 3) and info["prototypes_gts"].shape == (P,)
    idx = out["proto_index"].cpu().numpy()
    for j in range(P):
        bi, a = divmod(int(idx[j]), 4)
        assert info["prototypes_filenames"][j] == f"c{bi}_{a}"
        assert int(info["prototypes_gts"][j]) == int(loader[bi]["target_AS"][a])
        assert np.array_equal(info["prototypes_src_imgs"][j], loader[bi]["cine"][a].numpy())
        assert np.array_equal(info["prototypes_occurrence_maps"][j], rec[bi][1][a, j].numpy())
        assert np.array_equal(info["prototypes_preds"][j], rec[bi][2][a].numpy())
        assert info["prototypes_similarity_to_src_ROIs"][j] == np.float32(1) - rec[bi][0][a, j].numpy()


def test_push_prototypes_ppnet_end_to_end():
    from protoasnet_amd.push import push_prototypes_ppnet

    m = synth_model(CFG_PPNET_BOTTLENECK).to(DEV).eval()
    loader = _loader((3, 96, 96), nb=4, seed=60)
    recorded = []
    with torch.no_grad():
        for s in loader:
            c, d = m.push_forward(s["cine"].to(DEV))
            recorded.append((c.cpu().numpy(), d.cpu().numpy(), s["target_AS"].numpy()))
    ref_d, ref_p, ref_i = oracle.push.ppnet_push_select(recorded, m.prototype_class_identity.numpy(), m.num_classes,
                                                        m.prototype_shape, 4, True)
    out = push_prototypes_ppnet(loader, m, class_specific=True, replace_prototypes=True, log=lambda *_: None)
    idx = out["proto_index"].cpu().numpy()
    W = recorded[0][1].shape[3]
    assert np.array_equal(idx[:, 0], ref_i[:, 0]) and np.array_equal(idx[:, 1], ref_i[:, 1] * W + ref_i[:, 2])
    assert np.array_equal(m.prototype_vectors.detach().cpu().numpy(), ref_p.astype(np.float32))


# ------------------------------------------------------------------------------------------------------------------------
# G4: the winners the REFERENCE'S OWN push loops chose when run in the build container (tests/golden/make_golden_push.py)
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["ximg_cs_abstain", "ximg_cs", "ximg_all", "xvid_cs_abstain"])
def test_xproto_push_kernel_vs_reference_run(golden, tag):
    """pasn_push_xproto_update fed the reference's per-batch push_forward outputs: index-, distance- and vector-exact winners,
    incl. a repeated batch (the later one must win: '<=', push_abs_revision.py:299) and a repeated image inside a batch."""
    g = golden("g4_push.npz")
    cs, ab, B, K = (int(v) for v in g[f"{tag}_cfg"])
    feats, dist, labels = g[f"{tag}_batch_feats"], g[f"{tag}_batch_dist"], g[f"{tag}_labels"]
    P, D = feats.shape[2], feats.shape[3]
    ident = oracle.heads.prototype_class_identity(P, K).numpy()
    d, idx, vec = _xproto_sweep([(feats[i], dist[i], labels[i]) for i in range(feats.shape[0])], ident, K, bool(cs), bool(ab), P, D, B)
    w = g[f"{tag}_winners"]
    assert idx.tolist() == (w[:, 0] * B + w[:, 1]).tolist()
    assert np.array_equal(vec.reshape(g[f"{tag}_prototypes_after"].shape), g[f"{tag}_prototypes_after"])
    assert np.array_equal(np.float32(1) - d, g[f"{tag}_pickle_prototypes_similarity_to_src_ROIs"].astype(np.float32))


@pytest.mark.parametrize("tag", ["ppnet_cs", "ppnet_all"])
def test_ppnet_push_kernel_vs_reference_run(golden, tag):
    """pasn_push_ppnet_update fed the reference's per-batch (conv_output, distances): (image, h, w) exact, the first of two
    identical batches wins (strict '<', push_ProtoPNet.py:210), patches bit-exact."""
    g = golden("g4_push.npz")
    cs, B, K = (int(v) for v in g[f"{tag}_cfg"])
    conv, dist, labels = g[f"{tag}_batch_conv"], g[f"{tag}_batch_dist"], g[f"{tag}_labels"]
    P, D, H, W = dist.shape[2], conv.shape[2], dist.shape[3], dist.shape[4]
    ident = oracle.heads.prototype_class_identity(P, K).numpy()
    d, idx, vec = _ppnet_sweep([(conv[i], dist[i], labels[i]) for i in range(conv.shape[0])], ident, K, bool(cs), P, D, H * W, B)
    assert idx[:, 0].tolist() == g[f"{tag}_rf_boxes"][:, 0].tolist()
    assert idx[:, 1].tolist() == [int(np.argmax(g[f"{tag}_self_act"][j])) for j in range(P)]
    assert np.array_equal(vec.reshape(P, D, 1, 1), g[f"{tag}_prototypes_after"])


def _safe(dist_all, winners_flat, masks, tol=2e-6):
    """Prototypes whose reference winner beats every other candidate by more than fp32 noise (the whole-model tests run the HIP
    trunk, whose distances differ from the reference's in the last bits; the kernel-level tests above are exact)."""
    ok = []
    for j in range(dist_all.shape[1]):
        c = dist_all[:, j][masks[j]]
        best = dist_all[winners_flat[j], j]
        ok.append(np.sum(np.abs(c - best) <= tol) == np.sum(c == best))
    return np.array(ok)


def test_push_prototypes_ximg_whole_model_vs_reference_run(golden, tmp_path):
    """The product's push_prototypes (HIP ResNet-18 trunk + head B + device sweep + pickle) over the fixture's loader against what
    the reference's push_prototypes did on the same loader and weights: winners, projected prototypes, every pickle entry."""
    import pickle

    from protoasnet_amd.push import push_prototypes
    from util import CFG_PUSH_XIMG, push_loader

    g = golden("g4_push.npz")
    tag = "ximg_cs_abstain"
    m = synth_model(CFG_PUSH_XIMG).to(DEV).eval()
    loader = push_loader("image", (3, 64, 64))
    out = push_prototypes(loader, m, class_specific=True, abstain_class=True, replace_prototypes=True, log=lambda *_: None,
                          root_dir_for_saving_prototypes=str(tmp_path), epoch_number=7)
    w = g[f"{tag}_winners"]
    B, P = 4, w.shape[0]
    want_idx = w[:, 0] * B + w[:, 1]
    dist_all = g[f"{tag}_batch_dist"].reshape(-1, P)
    labels = g[f"{tag}_labels"].reshape(-1)
    cls = np.arange(P) // (P // 4)
    masks = [(labels == cls[j]) if cls[j] < 3 else np.ones_like(labels, bool) for j in range(P)]
    # an exact tie (the repeated batch) is decided by the rule, not by noise: count as safe when the tied set is the repeated pair
    safe = _safe(dist_all, want_idx, masks)
    idx = out["proto_index"].cpu().numpy()
    assert safe.sum() >= P - 2, "fixture should separate almost all winners"
    assert idx[safe].tolist() == want_idx[safe].tolist()
    got = m.prototype_vectors.detach().cpu().numpy()
    ref = g[f"{tag}_prototypes_after"]
    scale = np.abs(ref).max()
    assert np.abs(got - ref)[safe].max() <= 1e-5 * scale
    assert np.abs(out["proto_dist"].cpu().numpy() - (1 - g[f"{tag}_pickle_prototypes_similarity_to_src_ROIs"]))[safe].max() <= 2e-6
    with open(tmp_path / "epoch-7" / "prototypes_info.pickle", "rb") as fh:
        info = pickle.load(fh)
    assert sorted(info) == list(g[f"{tag}_pickle_keys"])
    assert [info["prototypes_filenames"][j] for j in np.nonzero(safe)[0]] == [f"b{w[j, 0]}_{w[j, 1]}" for j in np.nonzero(safe)[0]]
    assert np.array_equal(info["prototypes_gts"][safe], g[f"{tag}_pickle_prototypes_gts"][safe])
    assert np.abs(info["prototypes_preds"] - g[f"{tag}_pickle_prototypes_preds"])[safe].max() <= 1e-5
    occ_ref = g[f"{tag}_pickle_prototypes_occurrence_maps"]
    assert info["prototypes_occurrence_maps"].shape == occ_ref.shape
    assert np.abs(info["prototypes_occurrence_maps"] - occ_ref)[safe].max() <= 1e-5 * max(1.0, np.abs(occ_ref).max())
    sums = info["prototypes_src_imgs"].astype(np.float64).reshape(P, -1).sum(1)
    assert np.allclose(sums[safe], g[f"{tag}_pickle_src_imgs_sum"][safe], rtol=0, atol=1e-6)


def test_push_prototypes_ppnet_whole_model_vs_reference_run(golden, tmp_path):
    """push_prototypes_ppnet (HIP trunk + head A + device sweep + box files) vs the reference's run: dataset image index, (h, w),
    projected prototypes, bb-receptive_field<epoch>.npy, bb<epoch>.npy and the self-activation maps (push_ProtoPNet.py:121-135)."""
    from protoasnet_amd.push import push_prototypes_ppnet
    from util import CFG_PUSH_PPNET, push_loader

    g = golden("g4_push.npz")
    tag = "ppnet_cs"
    m = synth_model(CFG_PUSH_PPNET).to(DEV).eval()
    loader = push_loader("image", (3, 64, 64))
    out = push_prototypes_ppnet(loader, m, class_specific=True, replace_prototypes=True, log=lambda *_: None,
                                root_dir_for_saving_prototypes=str(tmp_path), epoch_number=3, proto_bound_boxes_filename_prefix="bb",
                                prototype_self_act_filename_prefix="self_act")
    dist = g[f"{tag}_batch_dist"]  # (nb, B, P, H, W)
    nb, B, P, H, W = dist.shape
    labels = g[f"{tag}_labels"].reshape(-1)
    flat = dist.transpose(0, 1, 3, 4, 2).reshape(nb * B * H * W, P)
    want_img = g[f"{tag}_rf_boxes"][:, 0]
    want_s = np.array([int(np.argmax(g[f"{tag}_self_act"][j])) for j in range(P)])
    cls = np.arange(P) // (P // 3)
    masks = [np.repeat(labels == cls[j], H * W) for j in range(P)]
    safe = _safe(flat, want_img * H * W + want_s, masks, tol=2e-5)
    assert safe.sum() >= P - 1
    idx = out["proto_index"].cpu().numpy()
    assert idx[safe, 0].tolist() == want_img[safe].tolist() and idx[safe, 1].tolist() == want_s[safe].tolist()
    ref = g[f"{tag}_prototypes_after"]
    assert np.abs(m.prototype_vectors.detach().cpu().numpy() - ref)[safe].max() <= 1e-5
    ep = tmp_path / "epoch-3"
    assert np.array_equal(np.load(ep / "bb-receptive_field3.npy")[safe], g[f"{tag}_rf_boxes"][safe])
    bb, bb_ref = np.load(ep / "bb3.npy"), g[f"{tag}_bound_boxes_torch_bicubic"]
    assert np.array_equal(bb[safe][:, [0, 5, 6, 7]], bb_ref[safe][:, [0, 5, 6, 7]])
    assert np.abs(bb[safe][:, 1:5] - bb_ref[safe][:, 1:5]).max() <= 1  # percentile threshold met marginally -> one pixel
    for j in np.nonzero(safe)[0]:
        assert np.allclose(np.load(ep / f"self_act{j}.npy"), g[f"{tag}_self_act"][j], atol=1e-3, rtol=1e-4)


@pytest.mark.timeout(900)
def test_cfg4_full_size_push_sweep_10k_clips_rechunk_invariant():
    """BASELINE config 4 at its full size under the test suite: push_prototypes over 10 000 clips of 3 x 16 x 224 x 224 (bf16, X3D-S, 30
    prototypes, class-specific mask) -- 313 batches of 32 -- and the same clip sequence again in 625 batches of 16.  The sequence re-uses 32
    resident clips (global clip g has the content of clip g % 32 and the label (g % 32 + g // 32) % 3), so every prototype sees thousands
    of exact ties: the reference's `<=` rule keeps the LAST one (push_abs_revision.py:299).  Size-independent properties: every
    prototype finds a clip of its own class, distances lie in [0, 1], the projected prototype reproduces its distance, and re-chunking
    the sweep changes neither the winning clip's content and label nor (beyond fp noise) its distance."""
    from protoasnet_amd.push import push_prototypes

    m = synth_model(CFG_VIDEO_X3D).to(DEV).eval().set_compute_dtype(torch.bfloat16)
    xs = synth.echo_clips((32, 3, 16, 224, 224)).to(DEV).to(torch.bfloat16)
    total = 10000

    class Loader:
        def __init__(self, bs):
            self.batch_size = bs

        def __len__(self):
            return (total + self.batch_size - 1) // self.batch_size

        def __iter__(self):
            for g0 in range(0, total, self.batch_size):
                g = torch.arange(g0, min(g0 + self.batch_size, total))
                yield {"cine": xs[(g % 32).to(DEV)], "target_AS": (g % 32 + g // 32) % 3, "filename": None}

    protos0 = m.prototype_vectors.detach().clone()
    with torch.no_grad():
        _, pd0, _, _ = m.push_forward(xs)  # (32, P) distances of the resident clips to the initial prototypes: every content meets every label
    want_d, want_c = pd0.min(dim=0)
    outs = {}
    for bs in (32, 16):
        with torch.no_grad():
            m.prototype_vectors.copy_(protos0)
        outs[bs] = push_prototypes(Loader(bs), m, class_specific=True, abstain_class=False, replace_prototypes=True, log=lambda *_: None)
    torch.cuda.synchronize()
    a, b = outs[32], outs[16]
    idx_a, idx_b = a["proto_index"].cpu(), b["proto_index"].cpu()
    assert int((idx_a >= 0).sum()) == 30 and int((idx_b >= 0).sum()) == 30
    da, db = a["proto_dist"].cpu(), b["proto_dist"].cpu()
    assert float(da.min()) >= 0.0 and float(da.max()) <= 1.0
    cls = torch.argmax(m.prototype_class_identity, dim=1)
    for idx in (idx_a, idx_b):
        labels = (idx % 32 + idx // 32) % 3
        assert torch.equal(labels, cls), "every winner carries its prototype's class"
    top2 = pd0.topk(2, dim=0, largest=False)[0]
    clear = ((top2[1] - top2[0]) > 2e-4).cpu()  # prototypes whose nearest resident clip is unambiguous
    assert int(clear.sum()) >= 5  # (synthetic echo clips resemble each other: many prototypes have two resident clips within 2e-4)
    assert torch.equal((idx_a % 32)[clear], want_c.cpu()[clear]), "the sweep's winner is not the nearest resident clip"
    # (a clip's distances move by ~1e-5 with the batch it rides in: the ragged last batch takes other tile / split shapes; bf16 forward noise is 1e-3)
    assert float((da - want_d.cpu()).abs().max()) <= 5e-5, "swept distance vs the per-clip minimum"
    assert torch.equal((idx_a % 32)[clear], (idx_b % 32)[clear]), "re-chunking changed a winning clip"
    assert float((da - db).abs().max()) <= 5e-5, float((da - db).abs().max())
    # the projected prototypes: the winner's own features, so its distance to itself is what the sweep recorded
    with torch.no_grad():
        feats, pdist, _, _ = m.push_forward(xs)
    j = torch.arange(30)
    again = pdist[(idx_b % 32).to(DEV), j.to(DEV)].cpu()
    assert float(again.max()) <= 2e-3, f"a projected prototype is {float(again.max()):.3g} away from its own source clip"
