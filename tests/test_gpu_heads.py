"""GPU: the two prototype layers through the C-ABI against the oracle and the reference's golden vectors."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from conftest import TOL_LOGITS, TOL_SIM, assert_close, assert_discriminates
from util import head_b_state, video_features

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rows(x, dtype):
    """logical (N,C,...) fp32 -> channels-last rows [N][S][Cp]."""
    from protoasnet_amd.plan import round_up

    n, c = x.shape[:2]
    s = int(np.prod(x.shape[2:]))
    cp = round_up(c, 8)
    rows = torch.zeros(n, s, cp, dtype=dtype, device=DEV)
    rows[:, :, :c] = x.reshape(n, c, s).transpose(1, 2).to(DEV).to(dtype)
    return rows, s, cp


# ------------------------------------------------------------------------------------------ head A
def _l2_head(z, protos, fcw, activation, dtype, want_argmin=True):
    from protoasnet_amd import _lib

    n, d = z.shape[:2]
    rows, s, dp = _rows(z, dtype)
    p, k = protos.shape[0], fcw.shape[0]
    dist = torch.empty(n, p, s, device=DEV)
    mind = torch.empty(n, p, device=DEV)
    amin = torch.empty(n, p, dtype=torch.int32, device=DEV)
    logits = torch.empty(n, k, device=DEV)
    pr, fw = protos.reshape(p, d).contiguous().to(DEV), fcw.contiguous().to(DEV)
    _lib.check(_lib.lib().pasn_l2_head_fwd(rows.data_ptr(), pr.data_ptr(), fw.data_ptr(), dist.data_ptr(), mind.data_ptr(),
                                           amin.data_ptr(), logits.data_ptr(), n, s, d, dp, p, k, _lib.dtype_code(dtype),
                                           0 if activation == "log" else 1, 1e-4, 0))
    torch.cuda.synchronize()
    return dist.cpu(), mind.cpu(), amin.cpu(), logits.cpu()


@pytest.mark.parametrize("tag,P,D", [("regular", 30, 512), ("bottleneck", 12, 128)])
def test_l2_head_matches_reference_golden(golden, tag, P, D):
    """Inputs = the reference's own conv_features; outputs vs the reference's distances / min / logits (fp32, 1e-3)."""
    from protoasnet_amd import synth

    g = golden("g1_ppnet_resnet18.npz")
    z = torch.from_numpy(g[f"{tag}_conv_features"])
    protos = torch.from_numpy(synth.synth_tensor("prototype_vectors", (P, D, 1, 1)))
    fcw = torch.from_numpy(synth.synth_tensor("last_layer.weight", (3, P)))
    dist, mind, amin, logits = _l2_head(z, protos, fcw, "log", torch.float32)
    assert_close(dist.view(2, P, 7, 7), g[f"{tag}_distances"], 1e-3, 0, "distances")
    assert_close(mind, g[f"{tag}_min_distances"], 1e-3, 0, "min_distances")
    assert_close(logits, g[f"{tag}_logits"], TOL_LOGITS, 0, "logits")
    ref_arg = torch.from_numpy(g[f"{tag}_distances"]).reshape(2, P, 49).argmin(dim=2)
    # bit-exact patch index, except where the reference's own top-2 gap is below fp32 reduction noise
    srt = torch.from_numpy(g[f"{tag}_distances"]).reshape(2, P, 49).sort(dim=2).values
    decided = (srt[..., 1] - srt[..., 0]) > 1e-4
    assert decided.float().mean() > 0.9
    assert torch.equal(amin.long()[decided], ref_arg[decided])
    if tag == "regular":
        _, _, _, lin = _l2_head(z, protos, fcw, "linear", torch.float32)
        assert_close(lin, g["regular_logits_linear"], 2e-3, 0, "linear logits")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 64, 5, 5), (2, 24, 9, 9), (5, 520, 1, 3), (1, 8, 8, 8)])
@pytest.mark.parametrize("P", [6, 33, 70])
def test_l2_head_vs_oracle_shapes(shape, P, dtype):
    """Ragged sizes: S not a multiple of 32, D not a multiple of 16, P across one/two/three MFMA tiles."""
    torch.manual_seed(11)
    n, d, h, w = shape
    z = torch.rand(shape)
    protos = torch.rand(P, d, 1, 1)
    fcw = torch.randn(3, P)
    zr = z.to(dtype).float()
    pr = protos.to(dtype).float()  # the kernel feeds prototypes to the MFMA in the compute dtype
    sd = {"prototype_vectors": pr, "ones": torch.ones_like(pr), "last_layer.weight": fcw}
    ref = oracle.heads.ppnet_head(sd, zr)
    dist, mind, amin, logits = _l2_head(z, protos, fcw, "log", dtype)
    tol = 1e-3 if dtype == torch.float32 else 5e-2
    assert_close(dist.view(n, P, h, w), ref["distances"], tol, tol, "distances")
    assert_close(mind, ref["min_distances"], tol, tol, "min")
    assert_close(logits, ref["logits"], tol * 5, tol, "logits")
    assert (dist >= 0).all()
    # argmin consistent with the kernel's own distance map (first index on ties)
    assert torch.equal(amin.long(), dist.argmin(dim=2))


def test_l2_head_tie_takes_first_index():
    z = torch.zeros(1, 16, 6, 6)
    z[0, :, 2, 3] = 0.5
    z[0, :, 4, 1] = 0.5  # two identical patches -> identical distances
    protos = torch.full((2, 16, 1, 1), 0.5)
    _, mind, amin, _ = _l2_head(z, protos, torch.ones(1, 2), "log", torch.float32)
    assert amin.tolist() == [[2 * 6 + 3, 2 * 6 + 3]] and float(mind.max()) == 0.0


# ------------------------------------------------------------------------------------------ head B
def _xproto_head(x, sd, P, D, K, dtype, mode=0, chain=False):
    """chain: pasn_xproto_chain_fwd (fragment-major weights, one launch for the convs + pooling) instead of pasn_xproto_head_fwd."""
    from protoasnet_amd import _lib
    from protoasnet_amd.plan import pack_conv_weight, round_up

    n, cb = x.shape[:2]
    rows, s, cbp = _rows(x, dtype)

    def pack(name, cin_p, bias=True):
        w, kc, r = pack_conv_weight(sd[name + ".weight"].to(DEV), cin_p, dtype)
        if chain:
            w = w.view(r // 32, 32, kc // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
        b = None
        if bias:
            b = torch.zeros(r, device=DEV)
            b[: sd[name + ".bias"].numel()] = sd[name + ".bias"].to(DEV)
        return w, b

    dp, hp, pp = round_up(D, 8), round_up(D // 2, 8), round_up(P, 8)
    a1, a1b = pack("add_on_layers.0", cbp)
    a2, a2b = pack("add_on_layers.2", dp)
    o1, o1b = pack("occurrence_module.0", cbp)
    o2, o2b = pack("occurrence_module.2", dp)
    o3, _ = pack("occurrence_module.4", hp, bias=False)
    d = _lib.XProtoDesc(N=n, S=s, Cb=cb, Cbp=cbp, D=D, Dp=dp, Hd=D // 2, Hp=hp, P=P, Pp=pp, K=K, mode=mode)
    lib = _lib.lib()
    code = _lib.dtype_code(dtype)
    if chain:
        assert lib.pasn_xproto_chain_supported(ctypes.byref(d), code) == 1
    ws = torch.empty(int(lib.pasn_xproto_chain_workspace_bytes(ctypes.byref(d)) if chain else lib.pasn_xproto_head_workspace_bytes(ctypes.byref(d), code)),
                     dtype=torch.uint8, device=DEV)
    occ = torch.full((n, P, s), float("nan"), device=DEV)
    feat = torch.full((n, P, D), float("nan"), device=DEV)
    sim = torch.full((n, P), float("nan"), device=DEV)
    logits = torch.full((n, K), float("nan"), device=DEV)
    protos = sd["prototype_vectors"].reshape(P, D).contiguous().to(DEV)
    fcw = sd["last_layer.weight"].contiguous().to(DEV)
    _lib.check((lib.pasn_xproto_chain_fwd if chain else lib.pasn_xproto_head_fwd)(rows.data_ptr(), a1.data_ptr(), a1b.data_ptr(), a2.data_ptr(), a2b.data_ptr(), o1.data_ptr(),
                                        o1b.data_ptr(), o2.data_ptr(), o2b.data_ptr(), o3.data_ptr(), protos.data_ptr(), fcw.data_ptr(),
                                        occ.data_ptr(), feat.data_ptr(), sim.data_ptr(), logits.data_ptr(), ws.data_ptr(),
                                        ctypes.byref(d), code, 0))
    torch.cuda.synchronize()
    return occ.cpu(), feat.cpu(), sim.cpu(), logits.cpu()


@pytest.mark.parametrize("tag", ["small", "refcfg", "p30"])
def test_xproto_head_matches_reference_golden(golden, tag):
    """fp32 kernels vs what the reference's Video_XProtoNet produced for the same features and weights (1e-3)."""
    g = golden("g3_video_head.npz")
    shape = tuple(int(v) for v in g[f"{tag}_shape"])
    P, K = (int(v) for v in g[f"{tag}_PK"])
    sd = head_b_state(shape[1], 256, P, K, video=True)
    x = video_features(shape, seed=1234 + shape[0])
    occ, feat, sim, logits = _xproto_head(x, sd, P, 256, K, torch.float32)
    assert_close(sim, g[f"{tag}_similarity"], TOL_SIM, 0, "similarity")
    assert_close(1 - sim, g[f"{tag}_proto_dist"], TOL_SIM, 0, "prototype distances")
    assert_close(logits, g[f"{tag}_logits"], TOL_LOGITS, 0, "logits")
    assert_discriminates(g[f"{tag}_similarity"], TOL_SIM, name="similarity")
    assert_discriminates(g[f"{tag}_logits"], TOL_LOGITS, name="logits")
    scale = float(np.abs(g[f"{tag}_features_extracted"]).max())
    assert_close(feat, g[f"{tag}_features_extracted"], 1e-5 * scale + 1e-3, 1e-4, "features_extracted")
    occ = occ.view((shape[0], P, 1) + shape[2:])
    if tag == "refcfg":
        assert_close(occ[:, :, :, ::2, ::3, ::3], g[f"{tag}_occurrence_map_sub"], 1e-3, 1e-4, "occurrence sample")
        assert_close(occ.double().sum(dim=(2, 3, 4, 5)), g[f"{tag}_occurrence_map_sum"], 5e-2, 1e-4, "occurrence sums")
    else:
        assert_close(occ, g[f"{tag}_occurrence_map"], 1e-3, 1e-4, "occurrence_map")


def test_xproto_head_image_matches_reference_golden(golden):
    """2-D head (XProtoNet, Cb = D = 512, S = 49) fed with the reference trunk features recomputed by the pinned oracle."""
    from protoasnet_amd import synth
    from util import CFG_XPROTO, synth_model

    g = golden("g2_xprotonet_resnet18.npz")
    sd = synth_model(CFG_XPROTO).state_dict()
    feats = oracle.backbones.resnet18_features(sd, "cnn_backbone.", synth.echo_clips((2, 3, 224, 224)))
    occ, feat, sim, logits = _xproto_head(feats, sd, 40, 512, 4, torch.float32)
    assert_close(sim, g["similarity"], TOL_SIM, 0, "similarity")
    assert_close(logits, g["logits"], TOL_LOGITS, 0, "logits")
    assert_close(occ.view(2, 40, 1, 7, 7), g["occurrence_map"], 1e-3, 1e-4, "occurrence_map")
    assert_close(feat, g["features_extracted"], 0.2, 1e-4, "features_extracted (|F| ~ 1e4)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(1, 24, 7, 16, 6, 3), (4, 192, 100, 64, 30, 3), (2, 40, 33, 48, 68, 4)])
def test_xproto_head_vs_oracle_shapes(cfg, dtype):
    """Ragged sizes (S, Cb, D/2, P not multiples of the tiles), P > 64 (two accumulator chunks), mode 1."""
    n, cb, s, D, P, K = cfg
    sd = head_b_state(cb, D, P, K, video=True)
    x = video_features((n, cb, 1, 1, s), seed=77)
    ref = oracle.heads.xproto_head(sd, x)
    occ, feat, sim, logits = _xproto_head(x, sd, P, D, K, dtype)
    tol = 1e-3 if dtype == torch.float32 else 4e-2
    fscale = float(ref["features_extracted"].abs().max())
    assert_close(occ.view(ref["occurrence_map"].shape), ref["occurrence_map"], tol * 3, tol, "occurrence_map")
    assert_close(feat, ref["features_extracted"], tol * fscale, tol, "features_extracted")
    assert_close(sim, ref["similarity"], tol, 0, "similarity")
    assert_close(logits, ref["logits"], tol * 10, 0, "logits")
    occ1, feat1, sim1, _ = _xproto_head(x, sd, P, D, K, dtype, mode=1)
    assert torch.equal(occ1, occ) and torch.isnan(feat1).all() and torch.isnan(sim1).all()  # mode 1 touches only occ
    # the path is reproducible bit for bit (fixed-order slab reduction, no atomics)
    occ2, feat2, sim2, logits2 = _xproto_head(x, sd, P, D, K, dtype)
    assert torch.equal(feat, feat2) and torch.equal(sim, sim2) and torch.equal(logits, logits2)


@pytest.mark.parametrize("cfg", [(32, 192, 784, 40, 4), (3, 192, 100, 30, 3), (2, 96, 209, 64, 4), (1, 40, 3, 7, 2), (2, 192, 3200, 40, 4),
                                 (5, 256, 1568, 40, 4), (2, 256, 97, 30, 3), (3, 200, 193, 64, 4)])
def test_xproto_chain_head(cfg):
    """Head B with the intermediate maps resident in LDS (pasn_xproto_chain_fwd, bf16, D = 256) against the oracle and against the
    seven-launch path on the same inputs: the headline shape (784 positions: 8 tiles of 98), ragged tiles (100 = one tile, 209 = 3 x 70 -
    1), a single short tile, P = 64 / P < 32 (one prototype tile), a narrow trunk (40 and 96 channels: k-steps beyond the data are zero
    fragments), 31 tiles per clip; the 256-channel instance (96-row tiles: the reference's own video shape 5 x 256 x 8 x 14 x 14, a tile
    of 97 = 96 + 1 positions, 200 channels); and the occurrence-map-only mode."""
    n, cb, s, P, K = cfg
    D = 256
    sd = head_b_state(cb, D, P, K, video=True)
    x = video_features((n, cb, 1, 1, s), seed=91)
    ref = oracle.heads.xproto_head(sd, x)
    occ, feat, sim, logits = _xproto_head(x, sd, P, D, K, torch.bfloat16, chain=True)
    tol = 4e-2
    fscale = float(ref["features_extracted"].abs().max())
    assert_close(occ.view(ref["occurrence_map"].shape), ref["occurrence_map"], tol * 3, tol, "occurrence_map")
    assert_close(feat, ref["features_extracted"], tol * fscale, tol, "features_extracted")
    assert_close(sim, ref["similarity"], tol, 0, "similarity")
    assert_close(logits, ref["logits"], tol * 10, 0, "logits")
    # the same rounding points as the separate launches: the occurrence map agrees bit for bit, the pooled features to fp32 summation order
    occ7, feat7, sim7, logits7 = _xproto_head(x, sd, P, D, K, torch.bfloat16)
    assert torch.equal(occ, occ7)
    assert_close(feat, feat7, 1e-5 * fscale, 1e-5, "features_extracted vs the seven-launch path")
    assert_close(sim, sim7, 1e-5, 0, "similarity vs the seven-launch path")
    assert_close(logits, logits7, 1e-4, 0, "logits vs the seven-launch path")
    occ1, feat1, sim1, _ = _xproto_head(x, sd, P, D, K, torch.bfloat16, mode=1, chain=True)
    assert torch.equal(occ1, occ) and torch.isnan(feat1).all() and torch.isnan(sim1).all()
    occ2, feat2, sim2, logits2 = _xproto_head(x, sd, P, D, K, torch.bfloat16, chain=True)
    assert torch.equal(feat, feat2) and torch.equal(sim, sim2) and torch.equal(logits, logits2)


def test_xproto_chain_head_routing():
    from protoasnet_amd import _lib

    lib = _lib.lib()
    mk = lambda **kw: _lib.XProtoDesc(**{**dict(N=2, S=784, Cb=192, Cbp=192, D=256, Dp=256, Hd=128, Hp=128, P=40, Pp=40, K=4, mode=0), **kw})
    ok = lambda d, dt: lib.pasn_xproto_chain_supported(ctypes.byref(d), _lib.dtype_code(dt))
    assert ok(mk(), torch.bfloat16) == 1 and ok(mk(mode=1), torch.bfloat16) == 1
    assert ok(mk(), torch.float32) == 0                         # fp32 keeps the seven-launch path
    assert ok(mk(D=512, Dp=512, Hd=256, Hp=256), torch.bfloat16) == 0   # image head (D = 512)
    assert ok(mk(Cb=256, Cbp=256), torch.bfloat16) == 1        # R(2+1)D-18[:-3]: 256 channels, the 96-row instance
    assert ok(mk(Cb=512, Cbp=512), torch.bfloat16) == 0        # ResNet-18 trunks: 512 channels do not fit the tile
    assert ok(mk(P=68, Pp=72), torch.bfloat16) == 0
