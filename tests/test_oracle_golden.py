"""CPU: the oracle restatement against the golden vectors produced by RUNNING THE REFERENCE
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import oracle
from protoasnet_amd import synth
from conftest import assert_close
from util import CFG_PPNET, CFG_PPNET_BOTTLENECK, CFG_XPROTO, head_b_state, synth_model, video_features

torch.set_num_threads(min(8, torch.get_num_threads()))

# fp32 on the same CPU, same torch ops, same op order -> only reduction-order noise is allowed
ATOL, RTOL = 2e-5, 2e-5


@pytest.mark.parametrize("tag,cfg", [("regular", CFG_PPNET), ("bottleneck", CFG_PPNET_BOTTLENECK)])
def test_ppnet_resnet18_matches_reference(golden, tag, cfg):
    g = golden("g1_ppnet_resnet18.npz")
    sd = synth_model(cfg).state_dict()
    assert sorted(sd.keys()) == list(g[f"{tag}_state_keys"])
    out = oracle.nets.ppnet_forward(sd, synth.echo_clips((2, 3, 224, 224)))
    assert_close(out["backbone_features"], g[f"{tag}_backbone_features"], ATOL, RTOL, "resnet18 features")
    assert_close(out["conv_features"], g[f"{tag}_conv_features"], ATOL, RTOL, "conv_features")
    assert_close(out["distances"], g[f"{tag}_distances"], 1e-4, RTOL, "distances")
    assert_close(out["min_distances"], g[f"{tag}_min_distances"], 1e-4, RTOL, "min_distances")
    assert_close(out["logits"], g[f"{tag}_logits"], ATOL, RTOL, "logits")


def test_ppnet_linear_activation(golden):
    g = golden("g1_ppnet_resnet18.npz")
    sd = synth_model(CFG_PPNET).state_dict()
    out = oracle.nets.ppnet_forward(sd, synth.echo_clips((2, 3, 224, 224)), activation="linear")
    assert_close(out["logits"], g["regular_logits_linear"], 1e-3, RTOL, "linear logits")


def test_xprotonet_resnet18_matches_reference(golden):
    g = golden("g2_xprotonet_resnet18.npz")
    m = synth_model(CFG_XPROTO)
    sd = m.state_dict()
    assert sorted(sd.keys()) == list(g["state_keys"])
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    out = oracle.nets.xprotonet_forward(sd, synth.echo_clips((2, 3, 224, 224)))
    assert_close(out["occurrence_map"], g["occurrence_map"], ATOL, RTOL, "occurrence_map")
    assert_close(out["features_extracted"], g["features_extracted"], 2e-2, 1e-4, "features_extracted")  # values reach 1e4
    assert_close(out["similarity"], g["similarity"], 1e-6, 0, "similarity")
    assert_close(out["proto_dist"], g["proto_dist"], 1e-6, 0, "1 - similarity")
    assert_close(out["logits"], g["logits"], 1e-5, 0, "logits")
    occ = oracle.nets.compute_occurence_map(sd, synth.echo_clips((2, 3, 224, 224)))
    assert_close(occ, g["occurrence_map"], ATOL, RTOL, "compute_occurence_map")


@pytest.mark.parametrize("tag", ["small", "refcfg", "p30"])
def test_video_head_matches_reference(golden, tag):
    g = golden("g3_video_head.npz")
    shape = tuple(int(v) for v in g[f"{tag}_shape"])
    P, K = (int(v) for v in g[f"{tag}_PK"])
    sd = head_b_state(shape[1], 256, P, K, video=True)
    x = video_features(shape, seed=1234 + shape[0])
    out = oracle.heads.xproto_head(sd, x)
    assert_close(out["logits"], g[f"{tag}_logits"], 1e-5, 0, "logits")
    assert_close(out["similarity"], g[f"{tag}_similarity"], 1e-6, 0, "similarity")
    assert_close(out["features_extracted"], g[f"{tag}_features_extracted"], 1e-3, 1e-5, "features_extracted")
    if tag == "refcfg":
        assert_close(out["occurrence_map"][:, :, :, ::2, ::3, ::3], g[f"{tag}_occurrence_map_sub"], ATOL, RTOL, "occ sample")
        assert_close(out["occurrence_map"].double().sum(dim=(2, 3, 4, 5)), g[f"{tag}_occurrence_map_sum"], 1e-2, 1e-5, "occ sums")
    else:
        assert_close(out["occurrence_map"], g[f"{tag}_occurrence_map"], ATOL, RTOL, "occurrence_map")
    # the contraction form used for big shapes is the same function
    out2 = oracle.heads.xproto_head(sd, x, contract=True)
    assert_close(out2["similarity"], out["similarity"], 1e-6, 0, "contract vs broadcast")


def test_constructor_semantics(golden):
    g = golden("g5_ctor.npz")
    assert_close(oracle.heads.prototype_class_identity(30, 3), g["ppnet_identity"], 0, 0, "identity 30/3")
    assert_close(oracle.heads.prototype_class_identity(40, 4), g["xproto_identity"], 0, 0, "identity 40/4")
    assert_close(oracle.heads.last_layer_init(oracle.heads.prototype_class_identity(30, 3), -0.5), g["ppnet_last_layer"], 0, 0, "ppnet fc")
    assert_close(oracle.heads.last_layer_init(oracle.heads.prototype_class_identity(40, 4), 0), g["xproto_last_layer"], 0, 0, "xproto fc")
    assert_close(oracle.heads.last_layer_init(oracle.heads.prototype_class_identity(40, 4), 0), g["video_last_layer"], 0, 0, "video fc")
    ks, st, pd = oracle.backbones.resnet18_conv_info()
    for img in (224, 112):
        rf = oracle.receptive_field.proto_layer_rf_info_v2(img, ks, st, pd, 1)
        assert np.allclose(np.array(rf, dtype=np.float64), g[f"ppnet_rf_{img}"])
    assert list(g["ppnet_rf_224"]) == [7, 32, 435, 0.5]
    assert float(g["ppnet_epsilon"]) == 1e-4
    assert float(g["video_addon_bias_absmax"]) == 0.0  # _initialize_weights zeroes conv biases (ProtoPNet.py:319-320)


def test_trunk_shapes_and_unpinned_trunks_run():
    """R(2+1)D and X3D are 'parity unpinned' (no reference arithmetic available): check the stated output shapes."""
    from util import CFG_VIDEO_R2P1D, CFG_VIDEO_X3D

    sd = synth_model(CFG_VIDEO_R2P1D).state_dict()
    y = oracle.backbones.trunk("resnet2p1d_18", sd, "cnn_backbone.", synth.echo_clips((1, 3, 8, 32, 32)))
    assert tuple(y.shape) == (1, 256, 2, 4, 4) == (1,) + oracle.backbones.trunk_out_shape("resnet2p1d_18", (3, 8, 32, 32))
    assert oracle.backbones.trunk_out_shape("resnet2p1d_18", (3, 32, 224, 224)) == (256, 8, 28, 28)  # resnet_features.py:311-313
    assert oracle.backbones.r2plus1d_midplanes(64, 64) == 144 and oracle.backbones.r2plus1d_midplanes(64, 128) == 230
    sd = synth_model(CFG_VIDEO_X3D).state_dict()
    y = oracle.backbones.trunk("x3d_s", sd, "cnn_backbone.", synth.echo_clips((1, 3, 4, 64, 64)))
    assert tuple(y.shape) == (1, 192, 4, 2, 2) == (1,) + oracle.backbones.trunk_out_shape("x3d_s", (3, 4, 64, 64))
    assert oracle.backbones.trunk_out_shape("x3d_s", (3, 16, 224, 224)) == (192, 16, 7, 7)
    assert [oracle.backbones.x3d_round_width(int(2.25 * w), 0.0625) for w in (24, 48, 96, 192)] == [8, 8, 16, 32]


# ---- train mode: the oracle's batch-statistics passes + autograd vs the REFERENCE run in train mode (tests/golden/make_golden_train.py)
def _train_sd(cfg):
    import torch

    from protoasnet_amd import model_builder, synth
    from train_cases import kink_sparse_

    m = model_builder.build(cfg)
    synth.load_synth(m)
    kink_sparse_(m)
    return {k: (v.detach().clone().requires_grad_() if v.is_floating_point() and "running" not in k and k != "ones" else v.detach().clone())
            for k, v in m.state_dict().items()}  # `ones` is a requires_grad=False Parameter in the reference (ProtoPNet.py:136)


def _check_train_golden(g, tag, sd, outputs, loss):
    import numpy as np
    import torch

    from train_cases import FULL_GRADS

    loss.backward()
    for name, t in outputs.items():
        ref = g[f"{tag}_{name}"]
        assert np.allclose(t.detach().numpy(), ref, rtol=1e-4, atol=1e-5 * float(np.abs(ref).max())), name
    names, stats = list(g[f"{tag}_grad_names"]), g[f"{tag}_grad_stats"]
    assert sorted(names) == sorted(n for n, v in sd.items() if getattr(v, "grad", None) is not None)
    for n, (gmax, gsum, gsq) in zip(names, stats):
        gr = sd[str(n)].grad.double()
        assert abs(float(gr.abs().max()) - gmax) <= 1e-3 * gmax + 1e-12, (n, float(gr.abs().max()), gmax)
        assert abs(float((gr * gr).sum()) - gsq) <= 2e-3 * gsq + 1e-20, (n, "sum of squares")
    for n in FULL_GRADS:
        key = f"{tag}_grad::{n}"
        if key in g:
            ref = g[key]
            assert np.allclose(sd[n].grad.numpy(), ref, rtol=0, atol=1e-4 * float(np.abs(ref).max())), n
    for key in g.files:
        if key.startswith(f"{tag}_buf::"):
            n = key.split("::", 1)[1]
            assert np.allclose(sd[n].numpy(), g[key], rtol=1e-5, atol=1e-6), n  # running estimates moved exactly like the reference's


def test_oracle_train_mode_xprotonet_vs_reference_golden(golden):
    from train_cases import SHAPE, loss_weights
    from util import CFG_XPROTO

    g = golden("g7_train_resnet18.npz")
    sd = _train_sd(dict(CFG_XPROTO, img_size=SHAPE[-1]))
    x = synth.echo_clips(SHAPE)
    out = oracle.nets.xprotonet_train_forward(sd, x, arch="resnet18")
    wl, ws, wo = loss_weights(SHAPE[0], 40, 4, tuple(out["occurrence_map"].shape[3:]))
    loss = (out["logits"] * wl).sum() + (out["similarity"] * ws).sum() + (out["occurrence_map"] * wo).sum()
    _check_train_golden(g, "xproto", sd, {"logits": out["logits"], "similarity": out["similarity"], "occurrence_map": out["occurrence_map"]}, loss)


def test_oracle_train_mode_ppnet_vs_reference_golden(golden):
    from train_cases import SHAPE, loss_weights
    from util import CFG_PPNET

    g = golden("g7_train_resnet18.npz")
    sd = _train_sd(dict(CFG_PPNET, img_size=SHAPE[-1]))
    x = synth.echo_clips(SHAPE)
    out = oracle.nets.ppnet_train_forward(sd, x, arch="resnet18")
    wl, wm, _ = loss_weights(SHAPE[0], 30, 3, (1, 1))
    loss = (out["logits"] * wl).sum() + (out["min_distances"] * wm).sum()
    _check_train_golden(g, "ppnet", sd, {"logits": out["logits"], "min_distances": out["min_distances"]}, loss)


def test_golden_tolerances_discriminate_between_samples(golden):
    """The gates used against the reference's golden outputs (conftest.TOL_*) must fail when two samples are swapped -- round 1's
    1e-3 did not (similarities of the two golden clips differ by 2e-3, logits by 4e-4 ... 1e-3)."""
    from conftest import TOL_LOGITS, TOL_SIM, assert_discriminates

    g1, g2, g3 = golden("g1_ppnet_resnet18.npz"), golden("g2_xprotonet_resnet18.npz"), golden("g3_video_head.npz")
    assert_discriminates(g2["similarity"], TOL_SIM, name="g2 similarity")
    assert_discriminates(g2["logits"], TOL_LOGITS, name="g2 logits")
    for tag in ("regular", "bottleneck"):
        assert_discriminates(g1[f"{tag}_logits"], TOL_LOGITS, name=f"g1 {tag} logits")
    for tag in ("small", "refcfg", "p30"):
        assert_discriminates(g3[f"{tag}_similarity"], TOL_SIM, name=f"g3 {tag} similarity")
        assert_discriminates(g3[f"{tag}_logits"], TOL_LOGITS, name=f"g3 {tag} logits")
    with pytest.raises(AssertionError):  # ... and round 1's gate indeed could not
        assert_discriminates(g1["regular_logits"], 1e-3, name="g1 logits at 1e-3")
