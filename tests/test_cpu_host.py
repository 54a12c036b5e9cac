"""CPU: host-side logic -- C-ABI export check, module surface, plan compiler / arena planner, push tie rules, merges."""
import os
import re

import numpy as np
import pytest
import torch

import oracle
from conftest import REPO
from util import CFG_PPNET, CFG_VIDEO_R2P1D, CFG_VIDEO_X3D, CFG_XPROTO, synth_model


# ------------------------------------------------------------------------------------------ C-ABI
def test_library_exports_every_declared_symbol():
    """Every function include/*.h declares is exported by the built .so and bound in _lib.SIGNATURES (no compute calls)."""
    from protoasnet_amd import _lib

    declared = set()
    inc = os.path.join(REPO, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            text = open(os.path.join(inc, fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            declared |= set(re.findall(r"\b(pasn_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 14
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.pasn_version() == 100
    assert lib.pasn_last_error() is not None


def test_header_is_plain_c(tmp_path):
    """include/protoasnet_amd.h is the boundary a C / cgo / JNI binding would compile: it must be valid C (round 3 shipped a
    declaration block inside a struct -- legal C++ member declarations, not C)."""
    import subprocess

    src = tmp_path / "t.c"
    src.write_text('#include "protoasnet_amd.h"\nint main(void) { return sizeof(pasn_conv_desc) == 25 * 4 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(REPO, "include"), str(src), "-o", str(tmp_path / "t")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([str(tmp_path / "t")]).returncode == 0


def test_missing_library_fails_loudly(monkeypatch):
    from protoasnet_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libprotoasnet_amd.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_library_path_override_is_read_at_import_and_still_fails_loudly():
    """PASN_LIB_PATH (A/B runs of two builds of the C-ABI library on one box) replaces the in-tree path; a wrong one is an error, not a fallback."""
    import subprocess
    import sys

    code = ("import protoasnet_amd._lib as L\n"
            "assert L.LIB_PATH == '/nonexistent/other.so', L.LIB_PATH\n"
            "try:\n    L.lib()\nexcept RuntimeError as e:\n    assert 'no CPU fallback' in str(e)\n    print('loud')\n")
    env = dict(os.environ, PASN_LIB_PATH="/nonexistent/other.so", PYTHONPATH=REPO)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "loud" in out.stdout, out.stderr[-500:]


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "protoasnet_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn


# ------------------------------------------------------------------------------------------ module surface
def test_model_builder_contract():
    from protoasnet_amd import model_builder

    m = model_builder.build(CFG_VIDEO_R2P1D)
    assert type(m).__name__ == "Video_XProtoNet" and m.prototype_shape == (40, 256, 1, 1, 1)
    assert m.proto_layer_rf_info is None and m.img_size == 112
    keys = list(m.state_dict().keys())
    assert keys[:2] == ["prototype_vectors", "ones"]
    for k in ("cnn_backbone.backbone.0.0.weight", "cnn_backbone.backbone.3.1.conv2.0.3.weight", "cnn_backbone.backbone.2.0.downsample.1.running_var",
              "add_on_layers.0.weight", "add_on_layers.2.bias", "occurrence_module.4.weight", "last_layer.weight"):
        assert k in keys, k
    assert "occurrence_module.4.bias" not in keys
    assert m.cnn_backbone.backbone[2][0].conv1[0][0].out_channels == 230  # midplanes of torchvision's BasicBlock
    assert not m.ones.requires_grad and m.prototype_vectors.requires_grad
    assert not isinstance(m.prototype_class_identity, torch.nn.Parameter) and "prototype_class_identity" not in keys
    with pytest.raises(KeyError):
        model_builder.build({k: v for k, v in CFG_VIDEO_R2P1D.items() if k != "checkpoint_path"})
    with pytest.raises(AssertionError):
        model_builder.build(dict(CFG_VIDEO_R2P1D, prototype_shape="(41, 256, 1, 1, 1)"))  # P % K != 0, ProtoPNet.py:332
    with pytest.raises(FileNotFoundError, match="resnet18-5c106cde.pth"):  # (tests/test_cpu_dropin.py covers the file being there)
        model_builder.build(dict(CFG_XPROTO, pretrained=True))


def test_constructor_semantics_match_reference(golden):
    g = golden("g5_ctor.npz")
    from protoasnet_amd import model_builder

    p = model_builder.build(CFG_PPNET)
    x = model_builder.build(CFG_XPROTO)
    v = model_builder.build(CFG_VIDEO_R2P1D)
    assert np.array_equal(p.prototype_class_identity.numpy(), g["ppnet_identity"])
    assert np.array_equal(p.last_layer.weight.detach().numpy(), g["ppnet_last_layer"])
    assert np.array_equal(x.last_layer.weight.detach().numpy(), g["xproto_last_layer"])
    assert np.array_equal(v.last_layer.weight.detach().numpy(), g["video_last_layer"])
    assert list(p.proto_layer_rf_info) == list(g["ppnet_rf_224"]) and list(x.proto_layer_rf_info) == list(g["xproto_rf_224"])
    assert list(model_builder.build(dict(CFG_PPNET, img_size=112)).proto_layer_rf_info) == list(g["ppnet_rf_112"])
    assert p.epsilon == float(g["ppnet_epsilon"]) and not hasattr(v, "epsilon")
    for n, q in v.named_parameters():
        if n.endswith("bias") and ("add_on" in n or "occurrence" in n):
            assert float(q.detach().abs().max()) == 0.0
    assert str(p.features) == "resnet18_features" and "RESNET2P1D" in str(v.cnn_backbone).upper()
    with pytest.raises(Exception, match="NOT implemented"):
        p.get_cnn_backbone_out_channels(torch.nn.Linear(2, 2))


def test_reference_class_surface_leftovers():
    """Attributes / methods the reference classes carry although nothing in the reference calls them from outside the models
    (XProtoNet.py:48-49,75-85; Video_XProtoNet.py:64-65,106-109; ProtoPNet.py:165-187)."""
    import torch.nn.functional as F

    from protoasnet_amd import model_builder, nets

    x = model_builder.build(CFG_XPROTO)
    v = model_builder.build(CFG_VIDEO_R2P1D)
    for m in (x, v):
        assert isinstance(m.cosine_similarity, torch.nn.CosineSimilarity) and m.cosine_similarity.dim == 2
        assert isinstance(m.om_softmax, torch.nn.Softmax) and m.om_softmax.dim == -1
        assert callable(m.get_occurence_map_absolute_val) and callable(m.get_occurence_map_softmaxed)
        assert not any(k.startswith(("cosine_similarity", "om_softmax")) for k in m.state_dict())  # parameter-free: checkpoint keys untouched
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.get_occurence_map_absolute_val(torch.zeros(1, 512 if m is x else 256, 2, 2, *((2,) if m is v else ())))
    # _weighted_l2_convolution against the reference's own formula (conv form) on the CPU
    g = torch.Generator().manual_seed(3)
    inp, filt, wts = torch.randn(2, 16, 5, 4, generator=g), torch.rand(6, 16, 1, 1, generator=g), torch.rand(6, 16, 1, 1, generator=g)
    ref = F.relu(F.conv2d(inp**2, wts) - 2 * F.conv2d(inp, filt * wts) + (filt**2 * wts).sum(dim=(1, 2, 3)).view(-1, 1, 1))
    got = nets.PPNet._weighted_l2_convolution(inp, filt, wts)
    assert got.shape == (2, 6, 5, 4) and float((got - ref).abs().max()) < 1e-5
    # with unit weights it is the plain squared distance of _l2_convolution (ProtoPNet.py:189-207)
    ones = torch.ones_like(filt)
    d = ((inp[:, None] - filt[None, :, :, :, :].expand(2, 6, 16, 1, 1)) ** 2).sum(2)
    assert float((nets.PPNet._weighted_l2_convolution(inp, filt, ones) - d).abs().max()) < 1e-4


def test_prune_prototypes():
    from protoasnet_amd import model_builder

    p = model_builder.build(CFG_PPNET)
    w = p.last_layer.weight.detach().clone()
    p.prune_prototypes([0, 5, 29])
    assert p.num_prototypes == 27 and tuple(p.prototype_vectors.shape) == (27, 512, 1, 1) and tuple(p.ones.shape) == (27, 512, 1, 1)
    keep = [i for i in range(30) if i not in (0, 5, 29)]
    assert torch.equal(p.last_layer.weight, w[:, keep]) and tuple(p.prototype_class_identity.shape) == (27, 3)


def test_receptive_field_box():
    from protoasnet_amd.receptive_field import compute_rf_prototype

    rf = [7, 32, 435, 0.5]
    for idx in ([0, 0, 0], [3, 6, 2], [1, 3, 3]):
        assert compute_rf_prototype(224, idx, rf) == oracle.receptive_field.rf_prototype(224, idx, rf)


# ------------------------------------------------------------------------------------------ plan compiler (no launches)
@pytest.mark.parametrize("cfg,shape", [(CFG_VIDEO_X3D, (2, 3, 16, 224, 224)), (CFG_VIDEO_R2P1D, (2, 3, 32, 112, 112)), (CFG_XPROTO, (8, 3, 224, 224))])
def test_plan_shapes_and_arena(cfg, shape):
    from protoasnet_amd.plan import PlanBuilder

    m = synth_model(cfg)
    trunk = m.cnn_backbone
    pb = PlanBuilder(torch.device("cpu"), torch.bfloat16, torch.float32)
    x_in = pb.input(shape)
    y = trunk.build_plan(pb, x_in)
    want = oracle.backbones.trunk_out_shape(cfg["base_architecture"], shape[1:])
    got = (y.C, y.T, y.H, y.W) if len(shape) == 5 else (y.C, y.H, y.W)
    assert got == want
    plan = pb.finish(x_in, y)
    # live ranges that overlap in time must not overlap in the arena
    bufs = [b for b in pb.bufs if not b.external]
    for i, a in enumerate(bufs):
        assert a.offset % 256 == 0
        for b in bufs[i + 1:]:
            if a.first <= b.last and b.first <= a.last:
                assert a.offset + a.nbytes <= b.offset or b.offset + b.nbytes <= a.offset, "live buffers overlap"
    assert plan.arena_bytes < (0.6 if cfg["base_architecture"] == "resnet18" else 0.35) * plan.naive_bytes
    n_ops = len(plan.ops)
    # x3d: fused stem 1 + 26 blocks x (expand, depthwise, project) + 4 shortcuts + the SE gates.  With the matrix-core stencil (default)
    # the 11 gates of the stride-1 SE blocks are stand-alone launches; with it off (PASN_DWMFMA=0) the 5 gates of the stages up to 128
    # channels ride in their VALU stencil launches (PASN_SE_FUSE_MAXC moves the boundary, PASN_NO_SE_FUSE=1: all stand-alone).  The 4
    # stride-2 SE blocks (first block of every stage: 54, 108, 216, 432 channels): they stay on the VALU stencil, the narrow ones
    # with their gate fused.  (Round 2's opt-in fused expand + depthwise launches and its stride-2 matrix-core stencil were retired in
    # round 3: none beat the default route; the measurements are in profiles/README.md.)
    fused = 0
    # project conv of block i + expand conv of block i+1 chained in one launch (bf16): the 10 pairs of stage 4
    # (+ the 4 of stage 3 with PASN_XPAIR_ALL=1)
    paired = 0 if os.environ.get("PASN_NO_XPAIR") == "1" or fused else (14 if os.environ.get("PASN_XPAIR_ALL") == "1" else 10)
    # round 4: the 6 project + expand pairs of the 432-channel stage in one launch each (x3d_pe.hip: weights streamed per row tile)
    paired += 0 if os.environ.get("PASN_NO_PE") == "1" else 6
    # ... and the three blocks without squeeze-excite of that stage as ONE launch each (x3d_edp.hip: the stencil launch disappears; the
    # pair launch that computed the block's expand conv becomes a single project conv with the gate in its prologue: same count)
    paired += 0 if os.environ.get("PASN_NO_EDP") == "1" else 3
    max_c = int(os.environ.get("PASN_SE_FUSE_MAXC", "128"))
    mfma = os.environ.get("PASN_DWMFMA", "1") != "0" and "PASN_DWMFMA_MAXW" not in os.environ
    mfma_s2 = False
    # (channels, stride-2 SE blocks, stride-1 SE blocks) per stage
    se_blocks = ((54, 1, 1), (108, 1, 2), (216, 1, 5), (432, 1, 3))
    if os.environ.get("PASN_NO_SE_FUSE") == "1" or fused:
        gates = 15
    else:
        gates = sum((s2 if (c > max_c or (mfma_s2 and c < 432)) else 0) + (s1 if (c > max_c or mfma) else 0) for c, s2, s1 in se_blocks)
    # round 3: the strided shortcut convs of stages 2 and 3 ride in their blocks' project-conv launches (pasn_conv3d_short_fwd), and the ten
    # gates of the 216- and 432-channel stages are computed in the project convs' prologues (pasn_conv3d_se_fwd / _pair_se_fwd) -- bf16 only
    short_fused = 0 if os.environ.get("PASN_NO_SHORTFUSE") == "1" else 2
    se_prologue = 0 if (os.environ.get("PASN_NO_SE_PROLOGUE") == "1" or os.environ.get("PASN_WS") == "0") else 10
    # (x3d_expdw.hip runs expand conv + stride-2 stencil of the first blocks of stages 2 and 3 as one launch; those blocks trade the
    # stencil-fused gate for a stand-alone one: same count)
    # ... and, at stride 1, the two blocks of the 56-wide stage (expand + stencil [+ gate] -> fused launch [+ gate])
    expdw_s1 = 0 if (os.environ.get("PASN_EXPDW") == "0" or os.environ.get("PASN_EXPDW_S1") == "0") else 1 if os.environ.get("PASN_EXPDW_S1") == "1" else 2
    assert n_ops == {"x3d_s": 1 + 26 * 3 - fused - paired + 4 + gates - short_fused - se_prologue - expdw_s1, "resnet2p1d_18": 2 + 6 * 4 + 2, "resnet18": 2 + 16 + 3}[cfg["base_architecture"]], n_ops


def test_packed_weight_layout():
    from protoasnet_amd.plan import fold_norm, pack_conv_weight

    w = torch.arange(2 * 3 * 1 * 2 * 2, dtype=torch.float32).reshape(2, 3, 1, 2, 2)
    p, kc, rows = pack_conv_weight(w, 8, torch.float32)
    assert (kc, rows) == (8, 128) and tuple(p.shape) == (128, 4, 8)
    assert p[1, 3, 2] == w[1, 2, 0, 1, 1] and p[1, 3, 3:].abs().sum() == 0 and p[2:].abs().sum() == 0
    p16, kc16, _ = pack_conv_weight(w, 8, torch.bfloat16)
    assert kc16 == 16 and p16.dtype == torch.bfloat16
    bn = torch.nn.BatchNorm3d(2).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.tensor([2.0, 0.5])); bn.bias.copy_(torch.tensor([1.0, -1.0]))
        bn.running_mean.copy_(torch.tensor([0.5, 0.25])); bn.running_var.copy_(torch.tensor([4.0, 0.25]))
    s, b = fold_norm(bn, None, 2, 8, "cpu")
    x = torch.randn(5, 2, 1, 1, 1)
    assert torch.allclose(bn(x).flatten(1), x.flatten(1) * s[:2] + b[:2], atol=1e-6) and s[2:].abs().sum() == 0


def test_channels_last_rows_zero_copy():
    from protoasnet_amd.plan import channels_last_rows, logical_view

    store = torch.randn(2, 3, 4, 5, 24)
    view = logical_view(store, 20, video=True)
    assert tuple(view.shape) == (2, 20, 3, 4, 5)
    rows, s, cp = channels_last_rows(view, torch.float32)
    assert (s, cp) == (60, 24) and rows.data_ptr() == store.data_ptr() and torch.equal(rows.reshape(store.shape), store)
    img = logical_view(torch.randn(2, 1, 4, 5, 16), 16, video=False)
    rows, s, cp = channels_last_rows(img, torch.float32)
    assert (s, cp) == (20, 16) and rows.data_ptr() == img.data_ptr()
    planar = torch.randn(2, 20, 3, 4, 5)
    rows, s, cp = channels_last_rows(planar, torch.float32)
    assert cp == 24 and torch.equal(rows[:, :, :20], planar.reshape(2, 20, 60).transpose(1, 2)) and rows[:, :, 20:].abs().sum() == 0


# ------------------------------------------------------------------------------------------ push rules (oracle = restated reference loops)
def test_xproto_push_tie_rules_by_hand():
    """`<=` across batches: a later batch wins an exact tie; np.argmin inside a batch: first index (push_abs_revision.py:299-300)."""
    ident = oracle.heads.prototype_class_identity(4, 2).numpy()  # prototypes 0,1 -> class 0; 2,3 -> class 1
    f = lambda v: np.full((3, 4, 2), v, np.float32)
    b0 = (f(1.0), np.array([[.5, .9, .3, .3], [.2, .9, .3, .3], [.2, .9, .1, .3]], np.float32), np.array([0, 0, 1]))
    b1 = (f(2.0), np.array([[.2, .9, .1, .3], [.7, .1, .1, .3], [.2, .9, .9, .3]], np.float32), np.array([0, 1, 1]))
    d, feats, where = oracle.push.xproto_push_select([b0, b1], ident, 2, class_specific=True, abstain_class=False)
    assert where == [(1, 0), (1, 0), (1, 1), (1, 1)]
    # p0: b0 min .2 @1 (first of two), b1 .2 @0 ties -> later batch.  p1: class-0 clips only -> .9 both, later batch, idx 0.
    # p2: class 1: b0 .1 @2, b1 min(.1@1,.9@2)=.1 -> later batch @1.  p3: .3 everywhere -> later batch, first class-1 index 1.
    assert d.tolist() == pytest.approx([.2, .9, .1, .3])
    assert all(np.all(v == 2.0) for v in feats)
    # abstain: the last P/num_classes prototypes ignore the label
    d, _, where = oracle.push.xproto_push_select([b0], oracle.heads.prototype_class_identity(4, 4).numpy(), 4, True, True)
    # p0: class-0 clips 0,1 -> .2 @1.  p1: the only class-1 clip is #2.  p2: class 2 never appears.  p3 (abstain): all clips, first .3
    assert where == [(0, 1), (0, 2), None, (0, 0)]


def test_ppnet_push_tie_rules_by_hand():
    """strict `<`: the first batch keeps an exact tie; flattened (n_c,h,w) argmin maps back through the class list."""
    ident = oracle.heads.prototype_class_identity(2, 2).numpy()
    conv = lambda v: np.full((2, 3, 2, 2), v, np.float32)
    d0 = np.full((2, 2, 2, 2), .5, np.float32); d0[1, 0, 1, 0] = .25; d0[0, 1, 0, 1] = .125
    d1 = np.full((2, 2, 2, 2), .5, np.float32); d1[0, 0, 0, 0] = .25; d1[1, 1, 1, 1] = .0625
    gd, patches, idx = oracle.push.ppnet_push_select([(conv(1.0), d0, np.array([1, 0])), (conv(2.0), d1, np.array([0, 1]))],
                                                     ident, 2, (2, 3, 1, 1), 2, class_specific=True)
    assert idx.tolist() == [[1, 1, 0], [3, 1, 1]]  # p0 keeps batch 0 (tie .25); p1: image 0 of batch 0 is class 1 but .5; batch 1 wins .0625
    assert gd.tolist() == [.25, .0625] and patches[0].max() == 1.0 and patches[1].max() == 2.0


def test_merge_rules_equal_single_sweep():
    from protoasnet_amd.push import merge_ppnet, merge_xproto, shard_batches

    rng = np.random.default_rng(1)
    P, D, B, nb, K = 12, 3, 4, 7, 3
    ident = oracle.heads.prototype_class_identity(P, K).numpy()
    batches = [(rng.standard_normal((B, P, D)).astype(np.float32), (np.round(rng.random((B, P)) * 4) / 4).astype(np.float32),
                rng.integers(0, K, B)) for _ in range(nb)]
    full_d, full_f, full_w = oracle.push.xproto_push_select(batches, ident, K, True, False)
    states = []
    for r in range(3):
        rg = shard_batches(nb, r, 3)
        d, f, w = oracle.push.xproto_push_select([batches[i] for i in rg], ident, K, True, False)
        idx = torch.tensor([-1 if x is None else (rg.start + x[0]) * B + x[1] for x in w])
        vec = torch.stack([torch.zeros(D) if v is None else torch.from_numpy(v) for v in f])
        states.append((torch.from_numpy(d).float(), idx, vec))
    d, idx, vec = merge_xproto(states)
    assert idx.tolist() == [w[0] * B + w[1] for w in full_w]
    assert torch.equal(vec, torch.stack([torch.from_numpy(v) for v in full_f]))
    assert [list(shard_batches(7, r, 3)) for r in range(3)] == [[0, 1, 2], [3, 4, 5], [6]]
    assert list(shard_batches(2, 3, 4)) == []
    # PPNet merge: earliest (image, s) keeps a tie
    a = (torch.tensor([.5, .25]), torch.tensor([[4, 1], [9, 0]]), torch.ones(2, 2))
    b = (torch.tensor([.5, .25]), torch.tensor([[2, 7], [9, 3]]), torch.zeros(2, 2))
    d, idx, vec = merge_ppnet([a, b])
    assert idx.tolist() == [[2, 7], [9, 0]] and vec.tolist() == [[0, 0], [1, 1]]


def test_grey_first_layer_identity_on_the_oracle():
    """Why the grey path is exact: a conv over three identical channels equals the conv of the one channel with the weights summed
    over the input channels (zero padding included), and the device normalisation x*a+b reproduces bin_to_norm."""
    import torch.nn.functional as F

    from protoasnet_amd.data import ECHO_MEAN, ECHO_STD, bin_to_norm, gray_to_gray3

    g = torch.Generator().manual_seed(3)
    u = torch.rand(2, 1, 4, 20, 20, generator=g)
    w = torch.randn(24, 3, 1, 3, 3, generator=g)
    x3 = torch.stack([gray_to_gray3(bin_to_norm(c)) for c in u])
    want = F.conv3d(x3, w, stride=(1, 2, 2), padding=(0, 1, 1))
    got = F.conv3d(u * (1 / ECHO_STD) + (-ECHO_MEAN / ECHO_STD), w.sum(1, keepdim=True), stride=(1, 2, 2), padding=(0, 1, 1))
    assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)
    assert tuple(gray_to_gray3(u[0]).shape) == (3, 4, 20, 20) and gray_to_gray3(u[0]).stride(0) == 0  # a view, like the reference's expand


def test_wide_buffer_stores_carry_no_register_soffset():
    """A 12- or 16-byte buffer store whose soffset is an SGPR gets no wait state from the compiler before a VALU write to its data registers
    (LLVM models the hazard for an immediate soffset only); gfx950 needs one -- round 5's dw_tz.hip stored a register pair the next instruction
    had already overwritten, in a few percent of the runs (profiles/README.md; tools/store_hazard_scan.py finds the pattern in assembly).
    The sources keep the frame offset in the VECTOR offset instead: every such store names soffset 0."""
    import glob
    import re

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "protoasnet_amd", "csrc")
    calls = 0
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        text = open(path).read()
        for m in re.finditer(r"__builtin_amdgcn_raw_buffer_store_b(?:96|128)\(", text):
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            args = text[m.end():i - 1]
            calls += 1
            assert re.search(r",\s*0\s*,\s*0\s*$", args), f"{os.path.basename(path)}: wide buffer store with a register soffset: {args[-80:]}"
    assert calls >= 8


def test_store_hazard_scanner_finds_the_pattern(tmp_path):
    """tools/store_hazard_scan.py on three snippets: the failing sequence of round 5 (16-byte buffer store with an SGPR soffset, its first data
    register overwritten by the next instruction), the same store behind an immediate soffset, and a store followed by an unrelated write."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("store_hazard_scan", os.path.join(root, "tools", "store_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = tmp_path / "bad.s"
    bad.write_text("\tbuffer_store_dwordx4 v[46:49], v50, s[12:15], s8 offen\n\tv_cndmask_b32_e32 v46, v72, v54, vcc\n")
    imm = tmp_path / "imm.s"
    imm.write_text("\tbuffer_store_dwordx4 v[46:49], v50, s[12:15], 0 offen\n\tv_cndmask_b32_e32 v46, v72, v54, vcc\n")
    other = tmp_path / "other.s"
    other.write_text("\tbuffer_store_dwordx4 v[46:49], v50, s[12:15], s8 offen\n\ts_add_i32 s8, s8, 1\n\tv_add_u32_e32 v50, s38, v54\n\tv_mov_b32_e32 v46, 0\n")
    assert mod.scan(str(bad)) == 1
    assert mod.scan(str(imm)) == 0
    assert mod.scan(str(other), window=2) == 0 and mod.scan(str(other), window=4) == 1
