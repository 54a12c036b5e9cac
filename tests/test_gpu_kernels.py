"""GPU: each trunk kernel, called through the C-ABI, against plain fp32 torch on the CPU for the same op.

fp32 kernels must agree to 1e-3 (BASELINE north_star tolerance; in practice ~1e-5); bf16 kernels are compared
with the fp32 result of the SAME bf16-rounded operands, so only accumulation / output rounding differs."""
import ctypes

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _lib_env(**kv):
    from protoasnet_amd import _lib

    return _lib.tuning_env(**kv)


def _pb(dtype, in_dtype=None):
    from protoasnet_amd.plan import PlanBuilder

    return PlanBuilder(torch.device(DEV), dtype, in_dtype or dtype)


def _run_single(pb, x_in, y_out, x_storage):
    plan = pb.finish(x_in, y_out)
    y = plan.run(x_storage)
    torch.cuda.synchronize()
    return y


def _to_cl(x, cp, dtype):
    """(N,C,T,H,W) fp32 -> channels-last storage [N][T][H][W][Cp] in dtype, zero padded."""
    n, c = x.shape[:2]
    out = torch.zeros((n,) + tuple(x.shape[2:]) + (cp,), dtype=dtype, device=DEV)
    out[..., :c] = x.permute(0, 2, 3, 4, 1).to(DEV).to(dtype)
    return out


def _from_cl(y, c):
    return y[..., :c].permute(0, 4, 1, 2, 3).float().cpu()


def _rt(x, dtype):
    """Round-trip through the compute dtype (what the kernel actually sees)."""
    return x.to(dtype).float()


def _tols(dtype):
    return (1e-3, 1e-4) if dtype == torch.float32 else (3e-2, 2e-2)


def _cl_input(pb, x, dtype):
    from protoasnet_amd.plan import Act, round_up

    n, c, t, h, w = x.shape
    cp = round_up(c, 8)
    store = _to_cl(x, cp, dtype)
    act = Act(n, t, h, w, c, cp, pb._new_buf(store.numel() * store.element_size(), external=True))
    return act, store


CONV_CASES = [
    # cin, cout, k, s, p, shape(T,H,W), act, residual
    (24, 54, (1, 1, 1), (1, 1, 1), (0, 0, 0), (3, 9, 7), "relu", False),     # X3D expand (Cout 54 -> pad 56)
    (54, 24, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 8, 8), "relu", True),      # X3D project + residual
    (24, 48, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 9, 9), "none", False),     # strided shortcut
    (45, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 6, 6), "relu", False),     # R(2+1)D stem temporal conv
    (64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 7, 7), "relu", False),    # Conv2Plus1D spatial
    (144, 64, (3, 1, 1), (2, 1, 1), (1, 0, 0), (5, 4, 4), "none", True),     # temporal, stride 2, residual
    (64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1), (4, 8, 8), "swish", False),   # generic 3x3x3
    (432, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 5, 5), "sigmoid", False),  # many k-steps, 6 output tiles (NT=3)
    (216, 216, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 6, 6), "abs", False),    # 7 output tiles (NT=4, 2 chunks)
    # pwconv_xtile instances (whole-K position tiles in LDS; S >= 64 positions per clip); 75 / 130 rows = ragged last tile
    (48, 216, (1, 1, 1), (1, 1, 1), (0, 0, 0), (3, 5, 5), "relu", False),    # KS = 4, two channel groups
    (96, 216, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 13, 5), "relu", False),   # KS = 6
    (216, 96, (1, 1, 1), (1, 1, 1), (0, 0, 0), (5, 4, 4), "relu", True),     # KS = 14 + residual, idle 4th wave
    (192, 432, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 7, 7), "none", False),   # KS = 12, four channel groups
    (432, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0), (4, 4, 4), "relu", True),    # KS = 28 (split weight loads) + residual
    (108, 48, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 8, 8), "none", True),     # KS = 8: preferred over the persistent kernel
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_mfma(case, dtype):
    cin, cout, k, s, p, thw, act, use_res = case
    torch.manual_seed(hash(case) % 1000)
    n = 2
    x = torch.randn(n, cin, *thw)
    conv = nn.Conv3d(cin, cout, k, s, p, bias=False)
    bn = nn.BatchNorm3d(cout)
    with torch.no_grad():
        conv.weight.mul_(2.0)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    xr = _rt(x, dtype)
    conv_r = nn.Conv3d(cin, cout, k, s, p, bias=False)
    conv_r.weight.data = _rt(conv.weight.data, dtype)
    ref = bn(conv_r(xr))
    res = torch.randn_like(ref) if use_res else None
    if res is not None:
        ref = ref + _rt(res, dtype)
    ref = {"relu": F.relu, "none": lambda v: v, "swish": lambda v: v * torch.sigmoid(v), "sigmoid": torch.sigmoid,
           "abs": torch.abs}[act](ref).detach()

    pb = _pb(dtype)
    xa, xs = _cl_input(pb, x, dtype)
    ra = rs = None
    if res is not None:
        ra, rs = _cl_input(pb, res, dtype)
    conv, bn = conv.to(DEV), bn.to(DEV)
    y = pb.conv(xa, conv, bn, act, residual=ra)
    plan = pb.finish(xa, y)
    if ra is not None:
        plan.ptrs[ra.buf] = rs.data_ptr()
    out = plan.run(xs)
    torch.cuda.synchronize()
    atol, rtol = _tols(dtype)
    assert_close(_from_cl(out, cout), ref, atol * max(1.0, float(ref.abs().max())), rtol, f"conv {case} {dtype}")
    if out.shape[-1] > cout:
        assert float(out[..., cout:].float().abs().max()) == 0.0, "padded channels must stay zero"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("gate", [False, True])
def test_conv3d_input_gate_swish(dtype, gate):
    """X3D project conv: x' = swish(x * gate[n][c]) fused into the operand load."""
    torch.manual_seed(3)
    n, cin, cout, thw = 3, 54, 24, (2, 5, 6)
    x = torch.randn(n, cin, *thw)
    g = torch.rand(n, cin) if gate else None
    conv = nn.Conv3d(cin, cout, 1, bias=False)
    xr = _rt(x, dtype)
    xin = xr * g[:, :, None, None, None] if gate else xr
    xin = _rt(xin * torch.sigmoid(xin), dtype)  # the kernel rounds the transformed operand to the MFMA input type
    ref = F.conv3d(xin, _rt(conv.weight.data, dtype)).detach()

    from protoasnet_amd.plan import round_up

    pb = _pb(dtype)
    xa, xs = _cl_input(pb, x, dtype)
    gbuf = gt = None
    if gate:
        gt = torch.zeros(n, round_up(cin, 8), dtype=torch.float32, device=DEV)
        gt[:, :cin] = g.to(DEV)
        gbuf = pb._new_buf(gt.numel() * 4, external=True)
    y = pb.conv(xa, conv.to(DEV), None, "none", in_gate=gbuf, in_swish=True)
    plan = pb.finish(xa, y)
    if gate:
        plan.ptrs[gbuf] = gt.data_ptr()
    out = plan.run(xs)
    torch.cuda.synchronize()
    atol, rtol = _tols(dtype)
    assert_close(_from_cl(out, cout), ref, atol * max(1.0, float(ref.abs().max())), rtol, "gate+swish conv")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout", [(216, 96), (432, 192), (96, 48)])
def test_pointwise_gate_swish_wide(cin, cout, dtype):
    """Wide project conv with the fused SE gate + Swish input transform (pwconv_xtile XF instances): three clips of 72
    positions, so a 64-row tile straddles two clips and must pick each row's own gate vector."""
    torch.manual_seed(cin)
    n, thw = 3, (2, 6, 6)
    x = torch.randn(n, cin, *thw)
    g = torch.rand(n, cin)
    conv = nn.Conv3d(cin, cout, 1, bias=False)
    res = torch.randn(n, cout, *thw)
    xr = _rt(x, dtype)
    xin = xr * g[:, :, None, None, None]
    xin = _rt(xin * torch.sigmoid(xin), dtype)
    ref = F.relu(F.conv3d(xin, _rt(conv.weight.data, dtype)) + _rt(res, dtype)).detach()

    from protoasnet_amd.plan import round_up

    pb = _pb(dtype)
    xa, xs = _cl_input(pb, x, dtype)
    ra, rs = _cl_input(pb, res, dtype)
    gt = torch.zeros(n, round_up(cin, 8), dtype=torch.float32, device=DEV)
    gt[:, :cin] = g.to(DEV)
    gbuf = pb._new_buf(gt.numel() * 4, external=True)
    y = pb.conv(xa, conv.to(DEV), None, "relu", residual=ra, in_gate=gbuf, in_swish=True)
    plan = pb.finish(xa, y)
    plan.ptrs[gbuf] = gt.data_ptr()
    plan.ptrs[ra.buf] = rs.data_ptr()
    out = plan.run(xs)
    torch.cuda.synchronize()
    atol, rtol = _tols(dtype)
    assert_close(_from_cl(out, cout), ref, atol * max(1.0, float(ref.abs().max())), rtol, f"wide gate+swish conv {cin}->{cout}")


def test_pointwise_gate_many_clips():
    """72 clips x 108 channels: the gate tensor (31 KB x ...) no longer fits the persistent kernel's 32 KB LDS budget, so
    the transform reads its gate rows from global memory instead (the other gate tests use the LDS-resident path)."""
    dtype = torch.bfloat16
    torch.manual_seed(5)
    n, cin, cout, thw = 80, 108, 48, (1, 3, 3)
    x = torch.randn(n, cin, *thw)
    g = torch.rand(n, cin)
    conv = nn.Conv3d(cin, cout, 1, bias=False)
    xr = _rt(x, dtype)
    xin = _rt((xr * g[:, :, None, None, None]) * torch.sigmoid(xr * g[:, :, None, None, None]), dtype)
    ref = F.conv3d(xin, _rt(conv.weight.data, dtype)).detach()
    from protoasnet_amd.plan import round_up

    pb = _pb(dtype)
    xa, xs = _cl_input(pb, x, dtype)
    gt = torch.zeros(n, round_up(cin, 8), dtype=torch.float32, device=DEV)
    gt[:, :cin] = g.to(DEV)
    assert gt.numel() * 4 > 32 * 1024
    gbuf = pb._new_buf(gt.numel() * 4, external=True)
    y = pb.conv(xa, conv.to(DEV), None, "none", in_gate=gbuf, in_swish=True)
    assert "persist" in pb.meta[-1]["kernel"]
    plan = pb.finish(xa, y)
    plan.ptrs[gbuf] = gt.data_ptr()
    out = plan.run(xs)
    torch.cuda.synchronize()
    atol, rtol = _tols(dtype)
    assert_close(_from_cl(out, cout), ref, atol * max(1.0, float(ref.abs().max())), rtol, "gate rows from global memory")


@pytest.mark.parametrize("kernel", ["ws", "xpair"])
@pytest.mark.parametrize("gate", [False, True])
@pytest.mark.parametrize("c0,c1,c2", [(216, 96, 216), (108, 48, 108)])
def test_conv_pair_chained(c0, c1, c2, gate, kernel, monkeypatch):
    """Project conv (+BN + residual + ReLU, optional SE gate + Swish on its input) chained with the next expand conv
    (+BN + ReLU) in one launch == the two torch convs; three clips of 72 positions (a tile straddles two clips), M not a
    multiple of 64."""
    monkeypatch.setenv("PASN_XPAIR_ALL", "1")  # also the narrow (stage-3) instance, off by default
    monkeypatch.setenv("PASN_WSPAIR", "2" if kernel == "ws" else "0")  # both implementations (2: every pair the persistent kernel covers, not only its default routes)
    dtype = torch.bfloat16
    torch.manual_seed(c0 + gate)
    n, thw = 3, (2, 6, 6)
    x = torch.randn(n, c0, *thw)
    res = torch.randn(n, c1, *thw)
    g = torch.rand(n, c0) if gate else None
    conv1, conv2 = nn.Conv3d(c0, c1, 1, bias=False), nn.Conv3d(c1, c2, 1, bias=False)
    bn1, bn2 = nn.BatchNorm3d(c1), nn.BatchNorm3d(c2)
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.3)
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.5, 1.5)
    bn1.eval(), bn2.eval()
    xin = _rt(x, dtype)
    if gate:
        xin = xin * g[:, :, None, None, None]
        xin = _rt(xin * torch.sigmoid(xin), dtype)
    y1 = F.relu(bn1(F.conv3d(xin, _rt(conv1.weight.data, dtype))) + _rt(res, dtype)).detach()
    y2 = F.relu(bn2(F.conv3d(_rt(y1, dtype), _rt(conv2.weight.data, dtype)))).detach()

    from protoasnet_amd.plan import round_up

    pb = _pb(dtype)
    xa, xs = _cl_input(pb, x, dtype)
    ra, rs = _cl_input(pb, res, dtype)
    gbuf = gt = None
    if gate:
        gt = torch.zeros(n, round_up(c0, 8), dtype=torch.float32, device=DEV)
        gt[:, :c0] = g.to(DEV)
        gbuf = pb._new_buf(gt.numel() * 4, external=True)
    pair = pb.conv_pair(xa, conv1.to(DEV), bn1.to(DEV), "relu", ra, conv2.to(DEV), bn2.to(DEV), "relu", in_gate=gbuf, in_swish=gate)
    assert pair is not None and ("pwconv_ws_kernel" if kernel == "ws" else "xpair") in pb.meta[-1]["kernel"], pb.meta[-1]["kernel"]
    o1, o2 = pair
    pb.bufs[o1.buf].external = True
    plan = pb.finish(xa, o2)
    out1 = torch.empty(n, *thw, o1.Cp, dtype=dtype, device=DEV)
    plan.ptrs[o1.buf] = out1.data_ptr()
    plan.ptrs[ra.buf] = rs.data_ptr()
    if gate:
        plan.ptrs[gbuf] = gt.data_ptr()
    out2 = plan.run(xs)
    torch.cuda.synchronize()
    atol, rtol = _tols(dtype)
    assert_close(_from_cl(out1, c1), y1, atol * max(1.0, float(y1.abs().max())), rtol, "chained pair: block output")
    assert_close(_from_cl(out2, c2), y2, atol * max(1.0, float(y2.abs().max())), rtol, "chained pair: expand output")
    for o, c in ((out1, c1), (out2, c2)):
        if o.shape[-1] > c:
            assert float(o[..., c:].float().abs().max()) == 0.0, "padded channels must stay zero"


WS_CASES = [
    # cin, cout, (N, T, H, W), residual, gate+swish, activation -- weight-stationary pointwise kernel (pwconv_ws.hip): every template k-step count,
    # both sub-tile counts, every (channel waves, position waves) split; M never a multiple of the block tile, clips shorter than two tiles so
    # that tiles straddle clips (two staged gate rows), blocks that walk several tiles (PASN_WS_BPC=1 + small tiles in the second pass)
    (48, 108, (3, 3, 9, 11), False, False, "relu"),    # KS 4, CT 4 x PT 2, MT 2, padded output channels (108 -> 112)
    (108, 48, (3, 3, 9, 11), True, True, "relu"),      # KS 8, CT 2 x PT 4, MT 1, gate + residual, half-empty second channel tile
    (108, 48, (3, 3, 9, 11), True, False, "relu"),     # ... without the input transform
    (96, 216, (2, 5, 7, 9), False, False, "relu"),     # KS 6, CT 7
    (216, 96, (3, 3, 8, 9), True, True, "relu"),       # KS 14, CT 3 x PT 2, MT 2
    (216, 96, (3, 3, 8, 9), True, False, "none"),
    (192, 432, (2, 4, 7, 7), False, False, "relu"),    # KS 12, 14 channel tiles in two groups of CT 7
    (432, 192, (3, 4, 7, 7), True, True, "relu"),      # KS 28, CT 6, MT 1
    (432, 192, (3, 4, 7, 7), True, False, "relu"),
    (192, 256, (2, 4, 7, 7), False, False, "relu"),    # head B: CT 8
    (256, 256, (2, 4, 7, 7), False, False, "none"),    # KS 16
    (256, 128, (2, 4, 7, 7), False, False, "relu"),    # CT 4 x PT 2
    (128, 30, (2, 8, 7, 7), False, False, "abs"),      # one channel tile: CT 1 x PT 8, 30 -> 32 padded channels
    (56, 24, (2, 4, 12, 12), True, True, "relu"),      # stage-2 shape class (Cin_p 56)
    (216, 96, (3, 3, 8, 9), False, True, "sigmoid"),   # Swish input without residual; sigmoid(0) != 0: the padded-channel mask must act
]


@pytest.mark.parametrize("shrink", [False, True])
@pytest.mark.parametrize("case", WS_CASES)
def test_pwconv_ws(case, shrink, monkeypatch):
    """Weight-stationary persistent pointwise kernel: against torch on the bf16-rounded operands AND bit-for-bit against the X-tile /
    persistent kernels it replaces (same MFMA k order, same epilogue arithmetic)."""
    cin, cout, (n, t, h, w), use_res, gate, act = case
    dtype = torch.bfloat16
    torch.manual_seed(cin * 7 + cout)
    x = torch.randn(n, cin, t, h, w)
    g = torch.rand(n, cin) + 0.25 if gate else None
    conv = nn.Conv3d(cin, cout, 1, bias=False)
    bn = nn.BatchNorm3d(cout)
    with torch.no_grad():
        conv.weight.mul_(2.0)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    xin = _rt(x, dtype)
    if gate:
        xin = xin * g[:, :, None, None, None]
        xin = _rt(xin * torch.sigmoid(xin), dtype)
    ref = bn(F.conv3d(xin, _rt(conv.weight.data, dtype)))
    res = torch.randn(n, cout, t, h, w) if use_res else None
    if res is not None:
        ref = ref + _rt(res, dtype)
    ref = {"relu": F.relu, "none": lambda v: v, "sigmoid": torch.sigmoid, "abs": torch.abs}[act](ref).detach()
    conv, bn = conv.to(DEV), bn.to(DEV)

    from protoasnet_amd.plan import round_up

    def run(ws: bool):
        monkeypatch.setenv("PASN_WS", "2" if ws else "0")  # 2: every layer the kernel covers, not only the ones it is routed to by default
        monkeypatch.setenv("PASN_WS_MINK", "48")
        if ws and shrink:  # one block per CU slot and the smallest tiles: blocks walk several tiles, the stage ring wraps
            monkeypatch.setenv("PASN_WS_BPC", "1")
            monkeypatch.setenv("PASN_WS_MT", "1")
            monkeypatch.setenv("PASN_WS_NS", "2")
        pb = _pb(dtype)
        xa, xs = _cl_input(pb, x, dtype)
        ra = rs = gbuf = gt = None
        if res is not None:
            ra, rs = _cl_input(pb, res, dtype)
        if gate:
            gt = torch.zeros(n, round_up(cin, 8), dtype=torch.float32, device=DEV)
            gt[:, :cin] = g.to(DEV)
            gbuf = pb._new_buf(gt.numel() * 4, external=True)
        y = pb.conv(xa, conv, bn, act, residual=ra, in_gate=gbuf, in_swish=gate)
        name = pb.meta[-1]["kernel"]
        plan = pb.finish(xa, y)
        if ra is not None:
            plan.ptrs[ra.buf] = rs.data_ptr()
        if gate:
            plan.ptrs[gbuf] = gt.data_ptr()
        out = plan.run(xs)
        torch.cuda.synchronize()
        for k in ("PASN_WS", "PASN_WS_MINK", "PASN_WS_BPC", "PASN_WS_MT", "PASN_WS_NS"):
            monkeypatch.delenv(k, raising=False)
        return out, name

    out, name = run(True)
    assert name.startswith("pwconv_ws_kernel"), name
    old, old_name = run(False)
    assert not old_name.startswith("pwconv_ws_kernel"), old_name
    scale = max(1.0, float(ref.abs().max()))
    assert_close(_from_cl(out, cout), ref, 3e-2 * scale, 2e-2, f"ws conv {case}")
    if "xtile" in old_name:  # identical arithmetic: the k order of the MFMA chain and the fp32 epilogue are the same
        assert torch.equal(out, old), f"ws vs {old_name}: max diff {float((out.float() - old.float()).abs().max())}"
    else:
        assert_close(_from_cl(out, cout), _from_cl(old, cout), 1.6e-2 * scale, 1e-2, f"ws vs {old_name} {case}")
    if out.shape[-1] > cout:
        assert float(out[..., cout:].float().abs().max()) == 0.0, "padded channels must stay zero"


@pytest.mark.parametrize("gate", [False, True])
@pytest.mark.parametrize("inner,cout,cin2,thw", [(54, 24, 24, (3, 11, 13)), (108, 48, 24, (2, 9, 10)), (108, 48, 48, (2, 8, 7))])
def test_conv_with_fused_strided_shortcut(inner, cout, cin2, thw, gate, monkeypatch):
    """First block of an X3D stage: project conv (+BN, optional SE gate + Swish on its input) with the block's strided 1x1x1 shortcut conv
    (+BN) accumulated in the same launch, then ReLU -- against torch on the bf16-rounded operands, and against the two separate launches
    (shortcut conv -> project conv with residual; the fused form skips the bf16 rounding of the shortcut tensor).  Odd planes (the last
    strided row / column exists), three clips, M not a multiple of the 32-row tile."""
    dtype = torch.bfloat16
    torch.manual_seed(inner + cin2 + gate)
    n = 3
    t, hi, wi = thw
    ho, wo = (hi - 1) // 2 + 1, (wi - 1) // 2 + 1
    z = torch.randn(n, inner, t, ho, wo)
    x2 = torch.randn(n, cin2, t, hi, wi)
    g = torch.rand(n, inner) + 0.25 if gate else None
    conv, conv2 = nn.Conv3d(inner, cout, 1, bias=False), nn.Conv3d(cin2, cout, 1, (1, 2, 2), bias=False)
    bn, bn2 = nn.BatchNorm3d(cout), nn.BatchNorm3d(cout)
    with torch.no_grad():
        for b in (bn, bn2):
            b.weight.uniform_(0.5, 1.5)
            b.bias.normal_(0, 0.3)
            b.running_mean.normal_(0, 0.3)
            b.running_var.uniform_(0.5, 1.5)
    bn.eval(), bn2.eval()
    zin = _rt(z, dtype)
    if gate:
        zin = zin * g[:, :, None, None, None]
        zin = _rt(zin * torch.sigmoid(zin), dtype)
    ref = F.relu(bn(F.conv3d(zin, _rt(conv.weight.data, dtype))) + bn2(F.conv3d(_rt(x2, dtype), _rt(conv2.weight.data, dtype), stride=(1, 2, 2)))).detach()
    conv, conv2, bn, bn2 = conv.to(DEV), conv2.to(DEV), bn.to(DEV), bn2.to(DEV)

    from protoasnet_amd.plan import round_up

    def run(fused: bool):
        pb = _pb(dtype)
        za, zs = _cl_input(pb, z, dtype)
        xa, xs = _cl_input(pb, x2, dtype)
        gbuf = gt = None
        if gate:
            gt = torch.zeros(n, round_up(inner, 8), dtype=torch.float32, device=DEV)
            gt[:, :inner] = g.to(DEV)
            gbuf = pb._new_buf(gt.numel() * 4, external=True)
        if fused:
            y = pb.conv_short(za, conv, bn, "relu", xa, conv2, bn2, in_gate=gbuf, in_swish=gate)
            assert y is not None and pb.meta[-1]["kind"] == "conv+shortcut", "the fused kernel must cover this pair"
        else:
            sc = pb.conv(xa, conv2, bn2, act="none")
            y = pb.conv(za, conv, bn, "relu", residual=sc, in_gate=gbuf, in_swish=gate)
        plan = pb.finish(za, y)
        plan.ptrs[xa.buf] = xs.data_ptr()
        if gate:
            plan.ptrs[gbuf] = gt.data_ptr()
        out = plan.run(zs)
        torch.cuda.synchronize()
        return out, len(plan.ops)

    out, n_fused = run(True)
    two, n_two = run(False)
    assert n_fused == 1 and n_two == 2
    scale = max(1.0, float(ref.abs().max()))
    assert_close(_from_cl(out, cout), ref, 3e-2 * scale, 2e-2, f"fused shortcut {inner}->{cout} + {cin2}->{cout}")
    assert_close(_from_cl(out, cout), _from_cl(two, cout), 1.6e-2 * scale, 1e-2, "fused vs the two launches")  # one bf16 ulp of the output
    if out.shape[-1] > cout:
        assert float(out[..., cout:].float().abs().max()) == 0.0, "padded channels must stay zero"


@pytest.mark.parametrize("se", [False, True])
@pytest.mark.parametrize("cin,cm,n,thw,stride", [(24, 54, 2, (4, 16, 30), 2), (24, 108, 3, (5, 13, 17), 2), (24, 54, 2, (18, 12, 56), 2),
                                                 (16, 40, 2, (3, 7, 9), 2), (24, 54, 1, (2, 2, 3), 2), (8, 72, 2, (3, 11, 16), 2),
                                                 (24, 54, 2, (4, 14, 56), 1), (24, 54, 2, (18, 7, 59), 1), (16, 40, 3, (3, 13, 70), 1),
                                                 (24, 54, 2, (5, 56, 56), 1), (24, 54, 1, (1, 9, 57), 1), (8, 24, 2, (7, 17, 71), 1)])
def test_expand_conv_and_strided_stencil_in_one_launch(cin, cm, n, thw, stride, se):
    """Front half of an X3D block: 1x1x1 expand conv + BN + ReLU -> depthwise 3x3x3 conv, stride (1,2,2) (a stage's first block) or stride 1
    (the blocks of planes >= 56 wide), + BN (+ Swish, or the squeeze-excite pool partial rows) in ONE launch with the expanded activation in LDS (pasn_x3d_expdw_fwd) -- against torch on the
    bf16-rounded operands and against the two launches.  Even and odd planes (the last strided row / column exists or not), planes
    smaller than a region, several regions and T chunks, two channel quads (108), block widths of 8 / 16 / 24 channels, channel counts
    that are not multiples of 16, T = 18 (chunked march), clips of 2 frames."""
    dtype = torch.bfloat16
    torch.manual_seed(cin + cm + n)
    t, hi, wi = thw
    x = torch.randn(n, cin, t, hi, wi)
    conv_a = nn.Conv3d(cin, cm, 1, bias=False)
    conv_b = nn.Conv3d(cm, cm, 3, (1, stride, stride), 1, groups=cm, bias=False)
    bn_a, bn_b = nn.BatchNorm3d(cm), nn.BatchNorm3d(cm)
    with torch.no_grad():
        conv_b.weight.mul_(3.0)
        for b in (bn_a, bn_b):
            b.weight.uniform_(0.5, 1.5)
            b.bias.normal_(0, 0.3)
            b.running_mean.normal_(0, 0.3)
            b.running_var.uniform_(0.5, 1.5)
    bn_a.eval(), bn_b.eval()
    act_b = "none" if se else "swish"
    e_ref = _rt(F.relu(bn_a(F.conv3d(_rt(x, dtype), _rt(conv_a.weight.data, dtype)))), dtype)
    pre = bn_b(F.conv3d(e_ref, _rt(conv_b.weight.data, dtype), stride=(1, stride, stride), padding=1, groups=cm)).detach()
    ref = pre if se else pre * torch.sigmoid(pre)
    conv_a, conv_b, bn_a, bn_b = conv_a.to(DEV), conv_b.to(DEV), bn_a.to(DEV), bn_b.to(DEV)

    def run(fused: bool, want_kernel: str = "x3d_expdw"):
        pb = _pb(dtype)
        xa, xs = _cl_input(pb, x, dtype)
        if fused:
            out = pb.expand_dw(xa, conv_a, bn_a, conv_b, bn_b, act_b, pool=se)
            assert out is not None and pb.meta[-1]["kind"] == "expand+dwconv", "the fused launch must cover this pair"
            assert pb.meta[-1]["kernel"].startswith(want_kernel + "<"), pb.meta[-1]["kernel"]
        else:
            e = pb.conv(xa, conv_a, bn_a, act="relu")
            out = pb.dwconv(e, conv_b, bn_b, act=act_b, pool=se)
        y, pooled = out if se else (out, None)
        if se:
            pb.bufs[pooled[0]].external = True
        plan = pb.finish(xa, y)
        pool_t = None
        if se:
            pool_t = torch.full((n, pooled[1], y.Cp), float("nan"), dtype=torch.float32, device=DEV)
            plan.ptrs[pooled[0]] = pool_t.data_ptr()
        o = plan.run(xs)
        torch.cuda.synchronize()
        return o, pool_t, len(plan.ops)

    # stride 1: the Toeplitz kernel on a channel-planar image (x3d_expdw_tz.hip, round 5) is the default; the block-diagonal kernel is the
    # PASN_EXPDW_TZ=0 route -- both are held to the same bounds, and to each other
    out, pool_f, n_f = run(True, "x3d_expdw_tz_kernel" if stride == 1 else "x3d_expdw_kernel")
    two, pool_t, n_t = run(False)
    assert n_f == 1 and n_t == 2
    scale = max(1.0, float(ref.abs().max()))
    if stride == 1:
        with _lib_env(PASN_EXPDW_TZ="0"):
            old, pool_o, _ = run(True, "x3d_expdw_kernel")
        assert_close(_from_cl(out, cm), _from_cl(old, cm), 1.6e-2 * scale, 1e-2, "Toeplitz vs block-diagonal fused launch")  # one bf16 ulp
        if se:  # (norm_b's scale meets the stencil weights before THEIR rounding in the Toeplitz kernel, after the MFMAs in the other: a per-channel
            # relative difference of up to one bf16 ulp per tap, which the sum over positions does not average away; both are held to the fp64 sums below)
            assert_close(pool_f.sum(1)[:, :cm], pool_o.sum(1)[:, :cm], 2e-2 * float(pre.abs().max()) * pre[0, 0].numel() ** 0.5 + 1e-3, 1e-2, "pool sums of the two fused kernels")
        for tc in (1, 3):  # chunked march (odd chunks: a step with one output frame) == the one-chunk march, bit for bit
            with _lib_env(PASN_EXPDW_TC=str(tc)):
                ch, pool_c, _ = run(True, "x3d_expdw_tz_kernel")
            assert torch.equal(ch, out), f"T chunk {tc}"
    assert_close(_from_cl(out, cm), ref, 3e-2 * scale, 2e-2, f"fused expand + stencil {cin}->{cm} {thw}")
    assert_close(_from_cl(out, cm), _from_cl(two, cm), 1.6e-2 * scale, 1e-2, "fused vs the two launches")  # one bf16 ulp of the output
    if out.shape[-1] > cm:
        assert float(out[..., cm:].float().abs().max()) == 0.0, "padded channels must stay zero"
    if se:  # the pool partial rows sum to the sum of the (pre-activation) output over positions
        want = pre.double().sum(dim=(2, 3, 4))
        got = pool_f.double().sum(dim=1)[:, :cm].cpu()
        assert_close(got, want, 2e-2 * float(pre.abs().max()) * pre[0, 0].numel() ** 0.5 + 1e-3, 1e-2, "pool partial sums")
        assert not torch.isnan(pool_f).any()


@pytest.mark.parametrize("c,cout,cse,n,thw,stride", [(432, 192, 32, 3, (4, 7, 7), 1), (432, 192, 32, 2, (5, 14, 14), 2), (216, 96, 16, 3, (4, 9, 9), 1)])
def test_project_conv_with_se_gate_in_its_prologue(c, cout, cse, n, thw, stride, monkeypatch):
    """X3D SE block back half: depthwise stencil (+BN, pool partial rows) -> project conv whose PROLOGUE computes the squeeze-excite gate from
    those rows (mean, fc1 + ReLU, fc2 + sigmoid), applies x' = swish(x * gate), adds the residual, ReLU -- one launch after the stencil --
    against torch, and against the stand-alone gate launch + gated conv.  Clips shorter than a block's row share and blocks that straddle
    two clips (three clips, ragged M)."""
    dtype = torch.bfloat16
    monkeypatch.setenv("PASN_WS", "2")  # the prologue lives in the weight-stationary kernel: take it for every covered layer
    torch.manual_seed(c + n)
    x = torch.randn(n, c, *thw)
    conv_b = nn.Conv3d(c, c, 3, (1, stride, stride), 1, groups=c, bias=False)
    conv_c = nn.Conv3d(c, cout, 1, bias=False)
    bn_b, bn_c = nn.BatchNorm3d(c), nn.BatchNorm3d(cout)
    fc1, fc2 = nn.Conv3d(c, cse, 1), nn.Conv3d(cse, c, 1)
    with torch.no_grad():
        for b in (bn_b, bn_c):
            b.weight.uniform_(0.5, 1.5)
            b.bias.normal_(0, 0.3)
            b.running_mean.normal_(0, 0.3)
            b.running_var.uniform_(0.5, 1.5)
    bn_b.eval(), bn_c.eval()
    conv_br = nn.Conv3d(c, c, 3, (1, stride, stride), 1, groups=c, bias=False)
    conv_br.weight.data = _rt(conv_b.weight.data, dtype)  # (the matrix-core stencil rounds its weights to bf16, the VALU one does not: loose gate below)
    yb = bn_b(conv_br(_rt(x, dtype))).detach()
    gate = torch.sigmoid(fc2(F.relu(fc1(yb.mean(dim=(2, 3, 4), keepdim=True))))).detach()
    yq = _rt(yb, dtype)
    zin = yq * gate
    zin = _rt(zin * torch.sigmoid(zin), dtype)
    res = torch.randn(n, cout, *yb.shape[2:])
    ref = F.relu(bn_c(F.conv3d(zin, _rt(conv_c.weight.data, dtype))) + _rt(res, dtype)).detach()
    mods = [m_.to(DEV) for m_ in (conv_b, bn_b, conv_c, bn_c, fc1, fc2)]

    def run(prologue: bool):
        monkeypatch.setenv("PASN_NO_SE_PROLOGUE", "0" if prologue else "1")
        pb = _pb(dtype)
        xa, xs = _cl_input(pb, x, dtype)
        ra, rs = _cl_input(pb, res, dtype)
        y, pooled = pb.dwconv(xa, mods[0], mods[1], act="none", pool=True)
        if prologue:
            out = pb.conv_se(y, mods[2], mods[3], "relu", ra, pooled, mods[4], mods[5])
            assert out is not None and pb.meta[-1]["kind"] == "conv+se", "the prologue kernel must cover this layer"
        else:
            g = pb.se_gate(pooled, mods[4], mods[5])
            out = pb.conv(y, mods[2], mods[3], "relu", residual=ra, in_gate=g, in_swish=True)
        plan = pb.finish(xa, out)
        plan.ptrs[ra.buf] = rs.data_ptr()
        o = plan.run(xs)
        torch.cuda.synchronize()
        return o, len(plan.ops)

    out, n1 = run(True)
    two, n2 = run(False)
    assert (n1, n2) == (2, 3)
    scale = max(1.0, float(ref.abs().max()))
    assert_close(_from_cl(out, cout), ref, 4e-2 * scale, 3e-2, f"project conv with SE prologue {c}->{cout}")
    assert_close(_from_cl(out, cout), _from_cl(two, cout), 1.6e-2 * scale, 1e-2, "prologue gate vs stand-alone gate launch")  # one bf16 ulp of the output


@pytest.mark.parametrize("chain", [True, False])
@pytest.mark.parametrize("n,t", [(3, 6), (2, 5), (5, 1), (2, 16)])
def test_x3d_whole_block_one_launch_7x7(n, t, chain, monkeypatch):
    """pasn_x3d_edp_fwd (x3d_edp.hip): a whole X3D block of the last stage -- expand conv + BN + ReLU -> depthwise 3x3x3 + BN + Swish -> project conv
    + BN + x + ReLU (-> the next block's expand conv + BN + ReLU) -- in ONE launch with both wide tensors in LDS: against torch with the
    rounding points of the separate launches, and BIT-IDENTICAL to those launches.  T even / odd (a tile with one output frame), T = 1 (every
    neighbour frame outside the clip), with and without the chained expand conv."""
    dtype = torch.bfloat16
    cx, cm = 192, 432
    torch.manual_seed(n * 10 + t)
    x = F.relu(torch.randn(n, cx, t, 7, 7))
    conv_a, conv_c, conv_n = nn.Conv3d(cx, cm, 1, bias=False), nn.Conv3d(cm, cx, 1, bias=False), nn.Conv3d(cx, cm, 1, bias=False)
    conv_b = nn.Conv3d(cm, cm, 3, 1, 1, groups=cm, bias=False)
    bns = [nn.BatchNorm3d(cm), nn.BatchNorm3d(cm), nn.BatchNorm3d(cx), nn.BatchNorm3d(cm)]
    with torch.no_grad():
        conv_a.weight.normal_(0, 0.1)
        conv_b.weight.normal_(0, 0.25)
        conv_c.weight.normal_(0, 0.08)
        conv_n.weight.normal_(0, 0.1)
        for b in bns:
            b.weight.uniform_(0.5, 1.5)
            b.bias.normal_(0, 0.3)
            b.running_mean.normal_(0, 0.3)
            b.running_var.uniform_(0.5, 1.5)
            b.eval()
    e = _rt(F.relu(bns[0](F.conv3d(_rt(x, dtype), _rt(conv_a.weight.data, dtype)))), dtype)
    d = bns[1](F.conv3d(e, _rt(conv_b.weight.data, dtype), padding=1, groups=cm))
    d = _rt(d * torch.sigmoid(d), dtype)
    y = _rt(F.relu(bns[2](F.conv3d(d, _rt(conv_c.weight.data, dtype))) + _rt(x, dtype)), dtype).detach()
    en = F.relu(bns[3](F.conv3d(y, _rt(conv_n.weight.data, dtype)))).detach()
    mods = [m_.to(DEV) for m_ in (conv_a, bns[0], conv_b, bns[1], conv_c, bns[2], conv_n, bns[3])]

    def run(whole: bool):
        with _lib_env(PASN_NO_EDP=None if whole else "1"):
            pb = _pb(dtype)
            xa, xs = _cl_input(pb, x, dtype)
            if whole:
                out = pb.x3d_edp(xa, *mods[:6], mods[6] if chain else None, mods[7] if chain else None)
                assert out is not None and pb.meta[-1]["kernel"].startswith("x3d_edp_kernel"), "the whole-block launch must cover this geometry"
                o1, o2 = out
            else:
                assert pb.x3d_edp(xa, *mods[:6], probe=True) is False
                ee = pb.conv(xa, mods[0], mods[1], act="relu")
                dd = pb.dwconv(ee, mods[2], mods[3], act="swish")
                o1 = pb.conv(dd, mods[4], mods[5], act="relu", residual=xa)
                o2 = pb.conv(o1, mods[6], mods[7], act="relu") if chain else None
            last = o2 if o2 is not None else o1
            if o2 is not None:
                pb.bufs[o1.buf].external = True
            plan = pb.finish(xa, last)
            out1 = torch.empty(n, t, 7, 7, o1.Cp, dtype=dtype, device=DEV)
            if o2 is not None:
                plan.ptrs[o1.buf] = out1.data_ptr()
            outl = plan.run(xs).clone()
            torch.cuda.synchronize()
            return ((out1, outl) if o2 is not None else (outl, None)) + (len(plan.ops),)

    f1, f2, nf = run(True)
    assert nf == 1
    assert_close(_from_cl(f1, cx), y, 4e-2 * max(1.0, float(y.abs().max())), 3e-2, "whole block: block output")
    if chain:
        assert_close(_from_cl(f2, cm), en, 4e-2 * max(1.0, float(en.abs().max())), 3e-2, "whole block: next expanded activation")
    u1, u2, nu = run(False)
    assert nu == (4 if chain else 3)
    assert torch.equal(f1, u1), "block output must be bit-identical to the separate launches"
    if chain:
        assert torch.equal(f2, u2), "expanded activation must be bit-identical to the separate launches"
    g1, g2, _ = run(True)
    assert torch.equal(f1, g1) and (not chain or torch.equal(f2, g2)), "bitwise reproducible"


@pytest.mark.parametrize("se", [False, True])
@pytest.mark.parametrize("n,thw", [(32, (4, 7, 7)), (3, (5, 7, 7)), (5, (2, 6, 5))])
def test_streamed_project_expand_pair_432(n, thw, se, monkeypatch):
    """pasn_x3d_pe_fwd (x3d_pe.hip): the 432 -> 192 -> 432 project + expand pair of X3D's last stage in ONE launch, weights streamed per row tile,
    with the block's squeeze-excite gate computed in the launch's prologue (se) -- against torch, and BIT-IDENTICAL to the launches it replaces
    (gate prologue + project conv on the weight-stationary kernel, expand conv on its own).  Shapes: tiles of 98 rows that never straddle a clip
    (the benchmark's layout, 32 clips), tiles that straddle clips (two gate rows per tile), a ragged last tile."""
    dtype = torch.bfloat16
    c, c1, c2, cse = 432, 192, 432, 32
    torch.manual_seed(n + thw[0] + se)
    x = torch.randn(n, c, *thw)
    res = torch.randn(n, c1, *thw)
    conv_b = nn.Conv3d(c, c, 3, 1, 1, groups=c, bias=False)
    conv_c, conv_a = nn.Conv3d(c, c1, 1, bias=False), nn.Conv3d(c1, c2, 1, bias=False)
    bn_b, bn_c, bn_a = nn.BatchNorm3d(c), nn.BatchNorm3d(c1), nn.BatchNorm3d(c2)
    fc1, fc2 = nn.Conv3d(c, cse, 1), nn.Conv3d(cse, c, 1)
    with torch.no_grad():
        conv_c.weight.normal_(0, 0.08)
        conv_a.weight.normal_(0, 0.1)
        for b in (bn_b, bn_c, bn_a):
            b.weight.uniform_(0.5, 1.5)
            b.bias.normal_(0, 0.3)
            b.running_mean.normal_(0, 0.3)
            b.running_var.uniform_(0.5, 1.5)
    for b in (bn_b, bn_c, bn_a):
        b.eval()
    conv_br = nn.Conv3d(c, c, 3, 1, 1, groups=c, bias=False)
    conv_br.weight.data = _rt(conv_b.weight.data, dtype)
    yb = bn_b(conv_br(_rt(x, dtype))).detach()
    if se:
        gate = torch.sigmoid(fc2(F.relu(fc1(yb.mean(dim=(2, 3, 4), keepdim=True))))).detach()
        zin = _rt(yb, dtype) * gate
        zin = _rt(zin * torch.sigmoid(zin), dtype)
    else:
        zin = _rt(yb * torch.sigmoid(yb), dtype)
    y1 = F.relu(bn_c(F.conv3d(zin, _rt(conv_c.weight.data, dtype))) + _rt(res, dtype)).detach()
    y2 = F.relu(bn_a(F.conv3d(_rt(y1, dtype), _rt(conv_a.weight.data, dtype)))).detach()
    mods = [m_.to(DEV) for m_ in (conv_b, bn_b, conv_c, bn_c, conv_a, bn_a, fc1, fc2)]

    def run(streamed: bool):
        with _lib_env(PASN_NO_PE=None if streamed else "1"):
            pb = _pb(dtype)
            xa, xs = _cl_input(pb, x, dtype)
            ra, rs = _cl_input(pb, res, dtype)
            if se:
                y, pooled = pb.dwconv(xa, mods[0], mods[1], act="none", pool=True)
                pair = pb.conv_pair(y, mods[2], mods[3], "relu", ra, mods[4], mods[5], "relu", in_swish=True, se=(pooled, mods[6], mods[7]))
            else:
                y = pb.dwconv(xa, mods[0], mods[1], act="swish")
                pair = pb.conv_pair(y, mods[2], mods[3], "relu", ra, mods[4], mods[5], "relu")
            if streamed:
                assert pair is not None and pb.meta[-1]["kernel"].startswith("x3d_pe_kernel"), "the streamed pair must cover this geometry"
                o1, o2 = pair
            else:
                assert pair is None, "without the streamed kernel nothing chains this width"
                o1 = pb.conv_se(y, mods[2], mods[3], "relu", ra, pooled, mods[6], mods[7]) if se else pb.conv(y, mods[2], mods[3], "relu", residual=ra)
                assert o1 is not None
                o2 = pb.conv(o1, mods[4], mods[5], "relu")
            pb.bufs[o1.buf].external = True
            plan = pb.finish(xa, o2)
            out1 = torch.empty(n, *thw, o1.Cp, dtype=dtype, device=DEV)
            plan.ptrs[o1.buf] = out1.data_ptr()
            plan.ptrs[ra.buf] = rs.data_ptr()
            out2 = plan.run(xs).clone()
            torch.cuda.synchronize()
            return out1, out2, len(plan.ops)

    f1, f2, nf = run(True)
    scale1, scale2 = max(1.0, float(y1.abs().max())), max(1.0, float(y2.abs().max()))
    assert_close(_from_cl(f1, c1), y1, 4e-2 * scale1, 3e-2, "streamed pair: block output")
    assert_close(_from_cl(f2, c2), y2, 4e-2 * scale2, 3e-2, "streamed pair: expand output")
    u1, u2, nu = run(False)
    assert (nf, nu) == (2, 3)
    assert torch.equal(f1, u1), "block output must be bit-identical to the separate launches"
    assert torch.equal(f2, u2), "expanded activation must be bit-identical to the separate launches"
    g1, g2, _ = run(True)
    assert torch.equal(f1, g1) and torch.equal(f2, g2), "bitwise reproducible"


@pytest.mark.parametrize("n,thw", [(3, (4, 9, 9)), (2, (3, 14, 14))])
def test_chained_pair_with_se_gate_in_its_prologue(n, thw, monkeypatch):
    """X3D stage-4 SE block followed by a block without shortcut: stencil (+ pool partial rows) -> ONE launch that computes the gate from
    those rows, runs the gated project conv (+BN + residual + ReLU) and the next block's expand conv (+BN + ReLU) from the block output
    kept on chip -- against the stand-alone gate launch + the chained pair with a gate tensor (pwconv_xpair), and against torch."""
    dtype = torch.bfloat16
    c, c1, c2, cse = 216, 96, 216, 16
    torch.manual_seed(n + thw[1])
    x = torch.randn(n, c, *thw)
    conv_b = nn.Conv3d(c, c, 3, 1, 1, groups=c, bias=False)
    conv_c, conv_a = nn.Conv3d(c, c1, 1, bias=False), nn.Conv3d(c1, c2, 1, bias=False)
    bn_b, bn_c, bn_a = nn.BatchNorm3d(c), nn.BatchNorm3d(c1), nn.BatchNorm3d(c2)
    fc1, fc2 = nn.Conv3d(c, cse, 1), nn.Conv3d(cse, c, 1)
    with torch.no_grad():
        for b in (bn_b, bn_c, bn_a):
            b.weight.uniform_(0.5, 1.5)
            b.bias.normal_(0, 0.3)
            b.running_mean.normal_(0, 0.3)
            b.running_var.uniform_(0.5, 1.5)
    for b in (bn_b, bn_c, bn_a):
        b.eval()
    conv_br = nn.Conv3d(c, c, 3, 1, 1, groups=c, bias=False)
    conv_br.weight.data = _rt(conv_b.weight.data, dtype)
    yb = bn_b(conv_br(_rt(x, dtype))).detach()
    gate = torch.sigmoid(fc2(F.relu(fc1(yb.mean(dim=(2, 3, 4), keepdim=True))))).detach()
    zin = _rt(yb, dtype) * gate
    zin = _rt(zin * torch.sigmoid(zin), dtype)
    res = torch.randn(n, c1, *thw)
    y1 = F.relu(bn_c(F.conv3d(zin, _rt(conv_c.weight.data, dtype))) + _rt(res, dtype)).detach()
    y2 = F.relu(bn_a(F.conv3d(_rt(y1, dtype), _rt(conv_a.weight.data, dtype)))).detach()
    mods = [m_.to(DEV) for m_ in (conv_b, bn_b, conv_c, bn_c, conv_a, bn_a, fc1, fc2)]

    def run(prologue: bool):
        pb = _pb(dtype)
        xa, xs = _cl_input(pb, x, dtype)
        ra, rs = _cl_input(pb, res, dtype)
        y, pooled = pb.dwconv(xa, mods[0], mods[1], act="none", pool=True)
        if prologue:
            pair = pb.conv_pair(y, mods[2], mods[3], "relu", ra, mods[4], mods[5], "relu", in_swish=True, se=(pooled, mods[6], mods[7]))
            assert pair is not None and pb.meta[-1]["kind"] == "conv_pair+se", "the prologue pair must cover this geometry"
        else:
            g = pb.se_gate(pooled, mods[6], mods[7])
            pair = pb.conv_pair(y, mods[2], mods[3], "relu", ra, mods[4], mods[5], "relu", in_gate=g, in_swish=True)
            assert pair is not None and "xpair" in pb.meta[-1]["kernel"], pb.meta[-1]["kernel"]
        o1, o2 = pair
        pb.bufs[o1.buf].external = True
        plan = pb.finish(xa, o2)
        out1 = torch.empty(n, *thw, o1.Cp, dtype=dtype, device=DEV)
        plan.ptrs[o1.buf] = out1.data_ptr()
        plan.ptrs[ra.buf] = rs.data_ptr()
        out2 = plan.run(xs)
        torch.cuda.synchronize()
        return out1, out2, len(plan.ops)

    a1, a2, na = run(True)
    b1, b2, nb = run(False)
    assert (na, nb) == (2, 3)
    for got, other, ref, cc, tag in ((a1, b1, y1, c1, "block output"), (a2, b2, y2, c2, "expand output")):
        scale = max(1.0, float(ref.abs().max()))
        assert_close(_from_cl(got, cc), ref, 4e-2 * scale, 3e-2, f"pair with SE prologue: {tag}")
        assert_close(_from_cl(got, cc), _from_cl(other, cc), 1.6e-2 * scale, 1e-2, f"prologue pair vs gate launch + chained pair: {tag}")


TCONV_CASES = [
    # cin, cout, (N, T, H, W), act, residual -- temporal (3,1,1) stride-1 convs whose weights fit a wave's registers (tconv_ws.hip)
    (144, 64, (2, 5, 9, 11), "relu", False),   # the 56 x 56 stage's layer: ragged last tile (99 positions), odd T
    (144, 64, (1, 32, 8, 8), "relu", True),    # a full march of 32 frames, one whole tile per frame, residual
    (144, 64, (3, 1, 7, 7), "none", True),     # T = 1: only the centre tap sees data
    (144, 64, (2, 2, 5, 29), "relu", True),    # T = 2, three tiles per frame
    (144, 45, (2, 4, 6, 6), "relu", False),    # 45 output channels: a channel tile with a tail
    (144, 24, (1, 6, 9, 9), "none", True),     # one channel tile (a block of two waves)
    (64, 64, (2, 7, 10, 13), "relu", True),    # 64 input channels (four k-steps per frame)
    (45, 64, (2, 5, 6, 7), "relu", False),     # R(2+1)D stem's temporal conv: 45 channels in 48 (three k-steps per frame, a zero-padded tail)
]


@pytest.mark.parametrize("case", TCONV_CASES)
def test_temporal_conv_weight_stationary(case, monkeypatch):
    """tconv_ws.hip (bf16): the (3,1,1) stride-1 conv as a weight-stationary pointwise conv over three frames of a T-marching LDS ring --
    against torch on the bf16-rounded operands, and against the halo-tile implicit GEMM it replaces (same products, fp32 accumulation in
    another order); repeated launches bit-identical."""
    cin, cout, nthw, act, use_res = case
    torch.manual_seed(sum(nthw) + cin + cout)
    dtype = torch.bfloat16
    x = torch.randn(*nthw[:1], cin, *nthw[1:])
    conv = nn.Conv3d(cin, cout, (3, 1, 1), 1, (1, 0, 0), bias=False)
    bn = nn.BatchNorm3d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    conv_r = nn.Conv3d(cin, cout, (3, 1, 1), 1, (1, 0, 0), bias=False)
    conv_r.weight.data = _rt(conv.weight.data, dtype)
    ref = bn(conv_r(_rt(x, dtype)))
    res = torch.randn_like(ref) if use_res else None
    if res is not None:
        ref = ref + _rt(res, dtype)
    ref = (F.relu(ref) if act == "relu" else ref).detach()
    conv, bn = conv.to(DEV), bn.to(DEV)

    def run(tconv):
        monkeypatch.setenv("PASN_TCONV", "1" if tconv else "0")
        pb = _pb(dtype)
        xa, xs = _cl_input(pb, x, dtype)
        ra = rs = None
        if res is not None:
            ra, rs = _cl_input(pb, res, dtype)
        y = pb.conv(xa, conv, bn, act, residual=ra)
        plan = pb.finish(xa, y)
        if ra is not None:
            plan.ptrs[ra.buf] = rs.data_ptr()
        out = plan.run(xs).clone()
        again = plan.run(xs).clone()
        torch.cuda.synchronize()
        assert torch.equal(out, again)
        return out, plan.meta[0]["kernel"]

    out, name = run(True)
    assert name == f"tconv_ws_kernel<{(cin + 15) // 16},{'true' if use_res else 'false'}>", name
    old, old_name = run(False)
    assert not old_name.startswith("tconv_ws"), old_name
    scale = max(1.0, float(ref.abs().max()))
    assert_close(_from_cl(out, cout), ref, 3e-2 * scale, 2e-2, f"temporal conv {case}")
    assert_close(_from_cl(out, cout), _from_cl(old, cout), 1.6e-2 * scale, 1e-2, "weight-stationary vs implicit GEMM")  # one bf16 ulp of the output
    if out.shape[-1] > cout:
        assert float(out[..., cout:].float().abs().max()) == 0.0, "padded channels must stay zero"


HALO_CASES = [
    # cin, cout, k, p, (N, T, H, W), act, residual -- stride-1 "same" convs of R(2+1)D-18 / ResNet-18 (igemm_halo.hip)
    (64, 144, (1, 3, 3), (0, 1, 1), (2, 3, 9, 11), "relu", False),    # spatial taps, 160-channel tile, tiles straddle rows / frames / clips
    (45, 64, (3, 1, 1), (1, 0, 0), (2, 5, 6, 7), "relu", False),      # temporal taps, kc 48 (half-empty second slice), ragged frames + positions
    (144, 64, (3, 1, 1), (1, 0, 0), (1, 9, 8, 8), "none", True),      # temporal taps, 4.5 channel slices, residual, T not a multiple of the box
    (128, 288, (1, 3, 3), (0, 1, 1), (1, 2, 14, 14), "relu", False),  # two 160-channel blocks, the second 128 wide
    (288, 128, (3, 1, 1), (1, 0, 0), (2, 4, 7, 7), "relu", True),     # 128-channel tile
    (64, 64, (1, 3, 3), (0, 1, 1), (3, 1, 14, 14), "relu", True),     # ResNet-18 BasicBlock conv (image = T 1)
    (256, 256, (1, 3, 3), (0, 1, 1), (2, 1, 7, 7), "none", False),    # 7 x 7: every position has a tap outside the image
    (64, 144, (1, 3, 3), (0, 1, 1), (1, 2, 40, 56), "relu", False),   # W = 56 (the halo of the stage-1 layers), several boxes
    (64, 64, (1, 3, 3), (0, 1, 1), (1, 1, 5, 112), "none", False),    # W = 112: the widest halo the tile takes (8 groups per wave)
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_igemm_halo_kernel(case, monkeypatch):
    """Halo-tile implicit GEMM (bf16, stride-1 windowed convs): against torch on the bf16-rounded operands, and against the per-tap
    kernel it replaces (same products, fp32 accumulation in a different order)."""
    from protoasnet_amd import _lib

    cin, cout, k, p, nthw, act, use_res = case
    torch.manual_seed(sum(nthw) + cin)
    dtype = torch.bfloat16
    n = nthw[0]
    x = torch.randn(n, cin, *nthw[1:])
    conv = nn.Conv3d(cin, cout, k, 1, p, bias=False)
    bn = nn.BatchNorm3d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    conv_r = nn.Conv3d(cin, cout, k, 1, p, bias=False)
    conv_r.weight.data = _rt(conv.weight.data, dtype)
    ref = bn(conv_r(_rt(x, dtype)))
    res = torch.randn_like(ref) if use_res else None
    if res is not None:
        ref = ref + _rt(res, dtype)
    ref = (F.relu(ref) if act == "relu" else ref).detach()
    conv, bn = conv.to(DEV), bn.to(DEV)

    monkeypatch.setenv("PASN_TCONV", "0")  # (the 144 -> 64 temporal layers take tconv_ws_kernel by default: test_temporal_conv_weight_stationary)

    def run(no_halo):
        monkeypatch.setenv("PASN_NO_HALO", "1" if no_halo else "0")
        pb = _pb(dtype)
        xa, xs = _cl_input(pb, x, dtype)
        ra = rs = None
        if res is not None:
            ra, rs = _cl_input(pb, res, dtype)
        y = pb.conv(xa, conv, bn, act, residual=ra)
        plan = pb.finish(xa, y)
        if ra is not None:
            plan.ptrs[ra.buf] = rs.data_ptr()
        out = plan.run(xs).clone()
        torch.cuda.synchronize()
        return out, plan.meta[0]["kernel"]

    out, name = run(False)
    assert name.startswith("igemm_halo_kernel"), name
    old, old_name = run(True)
    assert old_name.startswith("igemm_glds_kernel"), old_name
    scale = max(1.0, float(ref.abs().max()))
    assert_close(_from_cl(out, cout), ref, 3e-2 * scale, 2e-2, f"halo conv {case}")
    assert_close(_from_cl(out, cout), _from_cl(old, cout), 1.6e-2 * scale, 1e-2, f"halo vs per-tap kernel {case}")  # one bf16 ulp of the output
    if out.shape[-1] > cout:
        assert float(out[..., cout:].float().abs().max()) == 0.0, "padded channels must stay zero"


def test_conv_kernel_routing():
    """The variant query names the instance the launch will use (bench.py / profiles key on it)."""
    from protoasnet_amd import _lib
    from protoasnet_amd._lib import ConvDesc

    lib = _lib.lib()

    def desc(cin, cout, s=784, n=32, in_swish=0):  # the benchmark batch: few-position layers have their own kernel (last lines)
        rup = lambda v, m: (v + m - 1) // m * m
        return ConvDesc(N=n, Ti=1, Hi=1, Wi=s, Cin=cin, Cin_p=rup(cin, 8), To=1, Ho=1, Wo=s, Cout=cout, Cout_p=rup(cout, 8),
                        kt=1, kh=1, kw=1, st=1, sh=1, sw=1, pt=0, ph=0, pw=0, act=0, in_swish=in_swish,
                        w_kc=rup(rup(cin, 8), 16), w_rows=rup(cout, 128), w_frag=0)

    bf16 = _lib.dtype_code(torch.bfloat16)
    v = lambda d, gate=0: int(lib.pasn_conv3d_variant(ctypes.byref(d), bf16, gate))
    assert v(desc(24, 54)) == 1000 + 2 * 10 + 2          # persistent: weights in registers
    assert v(desc(54, 24)) == 1000 + 4 * 10 + 1
    assert v(desc(216, 96)) == 2500 + 2 * 14              # x-tile, KS = 14
    assert v(desc(216, 96, in_swish=1), 1) == 2500 + 2 * 14 + 1
    assert v(desc(432, 192)) == 2500 + 2 * 28
    assert v(desc(48, 108)) == 2500 + 2 * 4               # untransformed, 4 channel tiles: x-tile preferred
    assert v(desc(108, 48, in_swish=1), 1) == 1000 + 8 * 10 + 2  # gated: persistent kernel
    assert v(desc(216, 96, s=16)) < 2500                  # fewer than 64 positions per clip: not the x-tile kernel
    # weight-stationary persistent kernel (flags: bit 0 gate, bit 1 residual): the layers it is routed to by measurement
    assert v(desc(432, 192), 2) == 7000 + 28 * 10 + 1     # project + residual
    assert v(desc(432, 192, in_swish=1), 3) == 7000 + 28 * 10 + 1  # ... gated
    assert v(desc(108, 48), 2) == 7000 + 8 * 10 + 1
    assert v(desc(108, 48, in_swish=1), 3) == 1000 + 8 * 10 + 2    # gated narrow project conv: stays on the persistent kernel
    assert v(desc(96, 432)) == 7000 + 6 * 10 + 2          # wide expand conv
    assert v(desc(192, 432)) == 2500 + 2 * 12             # stays on the x-tile kernel
    # few positions (<= 8192): one wave per 32 x 32 tile straight from global memory -- fp32 from 64 input channels, bf16 from 256 (the image heads)
    f32 = _lib.dtype_code(torch.float32)
    assert int(lib.pasn_conv3d_variant(ctypes.byref(desc(512, 512, s=49, n=8)), f32, 0)) == 2002
    assert v(desc(512, 512, s=49, n=8)) == 2002 and v(desc(432, 192, n=2)) == 2002
    assert v(desc(192, 256, s=49, n=8)) != 2002            # bf16 below 256 input channels: the tiled kernels


FIRST_CASES = [
    (24, (1, 3, 3), (1, 2, 2), (0, 1, 1), (3, 17, 19), "none"),   # X3D stem conv_xy
    (45, (1, 7, 7), (1, 2, 2), (0, 3, 3), (2, 20, 22), "relu"),   # R(2+1)D stem
    (64, (1, 7, 7), (1, 2, 2), (0, 3, 3), (1, 30, 30), "relu"),   # ResNet-18 conv1 (image = T 1)
]


@pytest.mark.parametrize("dtypes", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("case", FIRST_CASES)
def test_first_conv(case, dtypes):
    cout, k, s, p, thw, act = case
    in_dtype, dtype = dtypes
    torch.manual_seed(5)
    n = 2
    x = torch.randn(n, 3, *thw)
    conv = nn.Conv3d(3, cout, k, s, p, bias=False)
    bn = nn.BatchNorm3d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    ref = bn(conv(_rt(x, in_dtype)))
    ref = (F.relu(ref) if act == "relu" else ref).detach()
    pb = _pb(dtype, in_dtype)
    xa = pb.input(tuple(x.shape))
    y = pb.first_conv(xa, conv.to(DEV), bn.to(DEV), act)
    out = _run_single(pb, xa, y, x.to(DEV).to(in_dtype).contiguous())
    atol, rtol = _tols(dtype)
    assert_close(_from_cl(out, cout), ref, atol * max(1.0, float(ref.abs().max())), rtol, f"first conv {case}")


FC_MFMA_CASES = [
    # cout, (kh, kw), pad, (T, H, W), act, grey -- stride (1,2,2) stems with W % 4 == 0 (first_conv_mfma.hip)
    (45, (7, 7), (3, 3), (2, 20, 24), "relu", False),    # R(2+1)D stem: 48-channel rows, second 32-channel tile half empty
    (64, (7, 7), (3, 3), (1, 36, 140), "relu", False),   # ResNet-18 conv1; 70 output columns = two column tiles, 18 rows = ragged row tile
    (45, (7, 7), (3, 3), (3, 16, 16), "none", True),     # grey clip: taps summed over the input channels, normalisation at load
    (24, (7, 7), (3, 3), (1, 12, 132), "relu", False),   # one 32-channel tile
    (24, (3, 3), (1, 1), (3, 18, 24), "none", False),    # X3D conv_xy (the unfused stem of the training path): window slot 3
    (24, (3, 3), (1, 1), (2, 9, 16), "relu", True),      # ... grey
]


@pytest.mark.parametrize("in_dtype", [torch.float32, torch.bfloat16, torch.uint8])
@pytest.mark.parametrize("case", FC_MFMA_CASES)
def test_first_conv_mfma(case, in_dtype, monkeypatch):
    """7x7 stride-2 stem on the matrix cores (bf16 out): against torch on the bf16-rounded operands, and against the fp32 VALU kernel."""
    cout, (kh, kw), (ph, pw), thw, act, grey = case
    if in_dtype == torch.uint8 and not grey:
        pytest.skip("uint8 clips enter through the grey pipeline")
    torch.manual_seed(cout + thw[2])
    n, cin = 2, 1 if grey else 3
    if in_dtype == torch.uint8:
        x8 = torch.randint(0, 256, (n, 1, *thw), dtype=torch.uint8)
        xin, mean, std, sc = x8, 0.099, 0.171, 255.0
        xf = (x8.float() / 255.0 - mean) / std
    else:
        xf = _rt(torch.randn(n, cin, *thw), in_dtype)
        xin, mean, std, sc = xf.to(in_dtype), 0.0, 1.0, 1.0
    conv = nn.Conv3d(3, cout, (1, kh, kw), (1, 2, 2), (0, ph, pw), bias=False)
    bn = nn.BatchNorm3d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    ref = bn(conv(xf.expand(n, 3, *thw) if grey else xf))
    ref = (F.relu(ref) if act == "relu" else ref).detach()

    def run(no_mfma):
        monkeypatch.setenv("PASN_NO_FC_MFMA", "1" if no_mfma else "0")
        pb = _pb(torch.bfloat16, in_dtype)
        if grey:
            pb.in_affine = (1.0 / (sc * std), -mean / std)
        xa = pb.input((n, cin, *thw))
        y = pb.first_conv(xa, conv.to(DEV), bn.to(DEV), act)
        out = _run_single(pb, xa, y, xin.to(DEV).contiguous()).clone()
        return out, pb.meta[-1]["kernel"]

    out, name = run(False)
    assert name.startswith("first_conv_mfma_kernel"), name
    old, old_name = run(True)
    assert old_name.startswith("first_conv_kernel"), old_name
    scale = max(1.0, float(ref.abs().max()))
    assert_close(_from_cl(out, cout), ref, 3e-2 * scale, 2e-2, f"mfma first conv {case} {in_dtype}")
    assert_close(_from_cl(out, cout), _from_cl(old, cout), 3e-2 * scale, 2e-2, f"mfma vs VALU first conv {case} {in_dtype}")
    if out.shape[-1] > cout:
        assert float(out[..., cout:].float().abs().max()) == 0.0, "padded channels must stay zero"


DW_CASES = [
    (24, (5, 1, 1), (1, 1, 1), (2, 0, 0), (6, 5, 5), "relu", False),    # X3D stem conv_t
    (54, (3, 3, 3), (1, 2, 2), (1, 1, 1), (3, 9, 9), "none", True),     # X3D conv_b stride 2 + SE pool
    (108, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 7, 6), "swish", False),  # X3D conv_b stride 1, Swish epilogue
    (432, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 13, 13), "none", True),  # widest stage, several pool blocks
    (520, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 5, 5), "none", True),    # > 512 channels: strip kernel + generic SE gate
    # the temporal kernel (dwtemporal.hip: a thread marches along T with the window in registers): T shorter than the window, a channel tail, kt = 3
    (24, (5, 1, 1), (1, 1, 1), (2, 0, 0), (1, 4, 7), "relu", False),
    (24, (5, 1, 1), (1, 1, 1), (2, 0, 0), (2, 3, 3), "none", False),
    (20, (5, 1, 1), (1, 1, 1), (2, 0, 0), (9, 6, 5), "swish", False),
    (45, (3, 1, 1), (1, 1, 1), (1, 0, 0), (7, 5, 6), "relu", False),
    (24, (5, 1, 1), (1, 1, 1), (2, 0, 0), (6, 4, 4), "none", True),     # with pool partial rows: the generic kernel keeps the layer
]


def _march_case(stride, c, dtype=torch.bfloat16, thw=None):
    torch.manual_seed(11)
    n = 2
    if thw is None:
        thw = (9, 11, 13) if stride == 1 else (9, 14, 22)
    x = torch.randn(n, c, *thw)
    conv = nn.Conv3d(c, c, 3, (1, stride, stride), 1, groups=c, bias=False)
    bn = nn.BatchNorm3d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    pre = bn(conv(_rt(x, dtype))).detach()
    return x, conv, bn, pre


def _run_march(x, conv, bn, act, dtype=torch.bfloat16):
    pb = _pb(dtype)
    xa, xs = _cl_input(pb, x, dtype)
    y, pooled = pb.dwconv(xa, conv.to(DEV), bn.to(DEV), act, pool=True)
    pool_buf, pool_blocks, _ = pooled
    pb.bufs[pool_buf].external = True
    plan = pb.finish(xa, y)
    part = torch.empty(x.shape[0], pool_blocks, y.Cp, dtype=torch.float32, device=DEV)
    plan.ptrs[pool_buf] = part.data_ptr()
    out = plan.run(xs)
    torch.cuda.synchronize()
    return out, part, pb.meta[-1]["kernel"]


@pytest.mark.parametrize("c", [24, 56, 108, 432])
@pytest.mark.parametrize("geom", ["0,0", "4,1", "4,3", "16,2", "2,1000"])
@pytest.mark.parametrize("thw", [(9, 11, 13), (9, 14, 22), (5, 7, 7), (4, 30, 8)])
def test_dwconv3d_mfma_variants(thw, geom, c, monkeypatch):
    """Matrix-core stencil (block-diagonal bf16 weight operands, LDS-DMA frame ring, T-marching accumulators): the cost model's own
    split and forced (T chunk, units per block) splits on ragged shapes -- T = 9 (chunk halos, partial last chunk), planes that are not
    multiples of the 4 x 14 region (ragged rows and strips), planes at most 8 wide (two output rows per position tile), channel counts
    with a partial 16-channel tile (24, 56, 108) and a short last channel quad (108, 432) -- with the Swish epilogue and SE partial
    sums, against torch; and against round 1's VALU stencil within the bf16 rounding of the weights (the only arithmetic difference:
    fp32 accumulation in both)."""
    tc, upb = geom.split(",")
    monkeypatch.setenv("PASN_DWMFMA", "1")  # every stride-1 layer (default: only the planes at most 8 wide)
    monkeypatch.setenv("PASN_DW_TZ", "0")   # (planes 9 .. 14 wide take the Toeplitz kernel by default)
    if tc != "0":
        monkeypatch.setenv("PASN_DWMFMA_TC", tc)
        monkeypatch.setenv("PASN_DWMFMA_UPB", upb)
    x, conv, bn, pre = _march_case(1, c, thw=thw)
    ref = pre * torch.sigmoid(pre)
    out, part, kernel = _run_march(x, conv, bn, "swish")
    assert kernel.startswith("dwconv3d_mfma_kernel<"), kernel
    atol, rtol = _tols(torch.bfloat16)
    assert_close(_from_cl(out, c), ref, atol * max(1.0, float(ref.abs().max())), rtol, f"mfma stencil {thw} {geom} c{c}")
    want = pre.sum(dim=(2, 3, 4))
    assert_close(part.sum(dim=1)[:, :c].cpu(), want, 2e-2 * float(want.abs().max()), 0, "SE partial sums")
    assert float(out[..., c:].abs().max() if out.shape[-1] > c else 0.0) == 0.0, "padded channels must stay zero"
    monkeypatch.setenv("PASN_DWMFMA", "0")  # (unset, the planes at most 8 wide take the matrix-core stencil by default)
    out1, part1, kernel1 = _run_march(x, conv, bn, "swish")
    assert kernel1.startswith("dwconv3d_march_kernel<"), kernel1
    d = (out.float() - out1.float()).abs()
    assert float(d.max()) <= 2.0 ** -6 * max(1.0, float(out1.float().abs().max())), float(d.max())


@pytest.mark.parametrize("act", ["swish", "none"])
@pytest.mark.parametrize("c", [24, 216])
@pytest.mark.parametrize("thw", [(16, 14, 14), (9, 13, 11), (1, 14, 14), (4, 9, 9), (3, 5, 14), (7, 14, 9)])
def test_dwconv3d_toeplitz_variants(thw, c, act, monkeypatch):
    """dw_tz.hip: the stride-1 stencil of planes 9 .. 14 wide in Toeplitz form on a channel-planar image (LDS-DMA of the channels-last rows +
    ds_read_b64_tr_b16; a block's two tiles are row bands of the whole plane): the 14 x 14 stage's plane, planes with a cut second band or
    none, odd T (a step with one output frame), T = 1, a partial 16-channel group.  With and without SE partial sums, against torch, against
    the block-diagonal matrix-core stencil (weights rounded before / after the norm's scale: one bf16 ulp of the output) and bit for bit
    across T chunks."""
    monkeypatch.setenv("PASN_DWMFMA", "1")
    x, conv, bn, pre = _march_case(1, c, thw=thw)
    ref = pre * torch.sigmoid(pre) if act == "swish" else pre
    out, part, kernel = _run_march(x, conv, bn, act)
    inst = f"{3 if act == 'swish' else 0},{{}}"
    assert kernel == f"dwconv3d_tz_kernel<{inst.format('true')}>", kernel
    atol, rtol = _tols(torch.bfloat16)
    scale = max(1.0, float(ref.abs().max()))
    assert_close(_from_cl(out, c), ref, atol * scale, rtol, f"toeplitz stencil {thw} c{c} {act}")
    want = pre.sum(dim=(2, 3, 4))
    assert_close(part.sum(dim=1)[:, :c].cpu(), want, 2e-2 * float(pre.abs().max()) * (pre[0, 0].numel() ** 0.5), 0, "SE partial sums")
    assert float(out[..., c:].abs().max() if out.shape[-1] > c else 0.0) == 0.0, "padded channels must stay zero"
    # without pool sums (norm's scale folded into the operands)
    pb = _pb(torch.bfloat16)
    xa, xs = _cl_input(pb, x, torch.bfloat16)
    y = pb.dwconv(xa, conv.to(DEV), bn.to(DEV), act)
    assert pb.meta[-1]["kernel"] == f"dwconv3d_tz_kernel<{inst.format('false')}>"
    out0 = pb.finish(xa, y).run(xs).clone()
    torch.cuda.synchronize()
    assert_close(_from_cl(out0, c), ref, atol * scale, rtol, f"toeplitz stencil, no pool {thw} c{c} {act}")
    for tc in (1, 3):
        monkeypatch.setenv("PASN_DWMFMA_TC", str(tc))
        outc, partc, kernelc = _run_march(x, conv, bn, act)
        assert kernelc == kernel and torch.equal(outc, out), f"T chunk {tc}"
        assert_close(partc.sum(dim=1)[:, :c].cpu(), part.sum(dim=1)[:, :c].cpu(), 1e-3 * float(pre.abs().max()) * (pre[0, 0].numel() ** 0.5), 0, "pool, T chunks")
    monkeypatch.delenv("PASN_DWMFMA_TC")
    monkeypatch.setenv("PASN_DW_TZ", "0")
    out1, part1, kernel1 = _run_march(x, conv, bn, act)
    assert kernel1.startswith("dwconv3d_mfma_kernel<"), kernel1
    dd = (out.float() - out1.float()).abs()
    assert float(dd.max()) <= 2.0 ** -6 * scale, float(dd.max())


@pytest.mark.parametrize("kernel", ["dw_tz", "expdw_tz"])
def test_toeplitz_kernels_repeat_bit_for_bit(kernel):
    """Both Toeplitz kernels at shapes that fill the card (several blocks per CU, full-length marches), launched 24 times on the same input:
    every run bit-identical to the first.  Round 5's store-data hazard (a 16-byte buffer store whose data register the next instruction
    overwrote: profiles/README.md entry 144) passed every single-run parity case and showed as one wrong output row in a few percent of the
    runs -- a repeat test is what sees that class of fault."""
    dtype = torch.bfloat16
    torch.manual_seed(5)
    pb = _pb(dtype)
    if kernel == "dw_tz":
        n, c, thw = 8, 216, (16, 14, 14)
        x = torch.randn(n, c, *thw)
        conv, bn = nn.Conv3d(c, c, 3, 1, 1, groups=c, bias=False).to(DEV), nn.BatchNorm3d(c).to(DEV).eval()
        xa, xs = _cl_input(pb, x, dtype)
        y = pb.dwconv(xa, conv, bn, "swish")
        assert pb.meta[-1]["kernel"].startswith("dwconv3d_tz_kernel<"), pb.meta[-1]["kernel"]
    else:
        n, cin, cm, thw = 4, 24, 54, (16, 56, 56)
        x = torch.randn(n, cin, *thw)
        mods = [nn.Conv3d(cin, cm, 1, bias=False), nn.BatchNorm3d(cm), nn.Conv3d(cm, cm, 3, 1, 1, groups=cm, bias=False), nn.BatchNorm3d(cm)]
        mods = [m.to(DEV).eval() for m in mods]
        xa, xs = _cl_input(pb, x, dtype)
        y = pb.expand_dw(xa, mods[0], mods[1], mods[2], mods[3], "swish", pool=False)
        assert y is not None and pb.meta[-1]["kernel"].startswith("x3d_expdw_tz_kernel<"), pb.meta[-1]["kernel"]
    plan = pb.finish(xa, y)
    first = plan.run(xs).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(first.float()).all()
    for rep in range(24):
        again = plan.run(xs)
        torch.cuda.synchronize()
        assert torch.equal(again, first), f"run {rep + 1} differs from the first in {int((again != first).sum())} elements"


@pytest.mark.parametrize("wt", [2, 3])
@pytest.mark.parametrize("tc", [4, 8, 16])
@pytest.mark.parametrize("stride", [1, 2])
def test_dwconv3d_march_variants(stride, tc, wt, monkeypatch):
    """T-marching stencil: every (outputs per strip, T chunk) instance on a shape with T = 9 (chunk halos, a partial last
    chunk), ragged W for both strip widths and SE partial sums; the cost model's own choice is covered by DW_CASES."""
    monkeypatch.setenv("PASN_DWMFMA", "0")  # the VALU stencil (stride-1 layers take the matrix-core one by default)
    monkeypatch.setenv("PASN_DWM_WT", str(wt))
    monkeypatch.setenv("PASN_DWM_TC", str(tc))
    dtype = torch.bfloat16
    torch.manual_seed(11)
    n, c, thw = 2, 56, (9, 11, 13) if stride == 1 else (9, 14, 22)
    x = torch.randn(n, c, *thw)
    conv = nn.Conv3d(c, c, 3, (1, stride, stride), 1, groups=c, bias=False)
    bn = nn.BatchNorm3d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    pre = bn(conv(_rt(x, dtype))).detach()
    ref = pre * torch.sigmoid(pre)
    pb = _pb(dtype)
    xa, xs = _cl_input(pb, x, dtype)
    y, pooled = pb.dwconv(xa, conv.to(DEV), bn.to(DEV), "swish", pool=True)
    assert "march" in pb.meta[-1]["kernel"] or stride == 2 and wt == 3  # stride 2 has no WT = 3 instance (falls back)
    pool_buf, pool_blocks, _ = pooled
    pb.bufs[pool_buf].external = True
    plan = pb.finish(xa, y)
    part = torch.empty(n, pool_blocks, y.Cp, dtype=torch.float32, device=DEV)
    plan.ptrs[pool_buf] = part.data_ptr()
    out = plan.run(xs)
    torch.cuda.synchronize()
    atol, rtol = _tols(dtype)
    assert_close(_from_cl(out, c), ref, atol * max(1.0, float(ref.abs().max())), rtol, f"march s{stride} tc{tc} wt{wt}")
    # SE partial sums are taken before the activation, over every output position
    got = part.sum(dim=1)[:, :c].cpu()
    want = pre.sum(dim=(2, 3, 4))
    assert_close(got, want, 2e-2 * float(want.abs().max()), 0, "SE partial sums")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", DW_CASES)
def test_dwconv3d_and_se(case, dtype):
    c, k, s, p, thw, act, pool = case
    torch.manual_seed(7)
    n = 2
    x = torch.randn(n, c, *thw)
    conv = nn.Conv3d(c, c, k, s, p, groups=c, bias=False)
    bn = nn.BatchNorm3d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    bn.eval()
    pre = bn(conv(_rt(x, dtype))).detach()
    ref = {"relu": F.relu, "none": lambda v: v, "swish": lambda v: v * torch.sigmoid(v)}[act](pre)

    pb = _pb(dtype)
    xa, xs = _cl_input(pb, x, dtype)
    conv, bn = conv.to(DEV), bn.to(DEV)
    atol, rtol = _tols(dtype)
    if not pool:
        y = pb.dwconv(xa, conv, bn, act)
        if k[1] == 1 and k[2] == 1:
            assert pb.meta[-1]["kernel"].startswith("dwconv_t_kernel<"), pb.meta[-1]["kernel"]
        out = _run_single(pb, xa, y, xs)
        assert_close(_from_cl(out, c), ref, atol * max(1.0, float(ref.abs().max())), rtol, f"dwconv {case}")
        return
    from protoasnet_amd.backbones import _SE

    se = _SE(c)
    with torch.no_grad():
        for q in se.parameters():
            q.normal_(0, 0.2)
    y, pooled = pb.dwconv(xa, conv, bn, act, pool=True)
    gate_buf = pb.se_gate(pooled, se.fc1.to(DEV), se.fc2.to(DEV))
    pb.bufs[gate_buf].external = True
    plan = pb.finish(xa, y)
    gate = torch.empty(n, y.Cp, dtype=torch.float32, device=DEV)
    plan.ptrs[gate_buf] = gate.data_ptr()
    out = plan.run(xs)
    torch.cuda.synchronize()
    assert_close(_from_cl(out, c), ref, atol * max(1.0, float(ref.abs().max())), rtol, f"dwconv {case}")
    se = se.cpu()
    gm = pre.mean(dim=(2, 3, 4), keepdim=True)
    gref = torch.sigmoid(se.fc2(F.relu(se.fc1(gm)))).reshape(n, c).detach()
    assert_close(gate[:, :c], gref, 1e-3 if dtype == torch.float32 else 1e-2, 0, "SE gate")


@pytest.mark.parametrize("dtypes", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("thw", [(7, 18, 22), (2, 9, 9), (1, 16, 16), (12, 11, 14)])
def test_x3d_stem_fused(thw, dtypes, monkeypatch):
    """Fused stem (spatial conv -> temporal depthwise conv -> BN -> ReLU, T-marching register ring) == torch, and
    bit-identical to the two unfused launches (ring values are rounded to the activation dtype exactly like the tensor
    the unfused path stores)."""
    in_dtype, dtype = dtypes
    monkeypatch.setenv("PASN_NO_STEM_MFMA", "1")  # the matrix-core stem (bf16, W % 4 == 0) is a tolerance path: test_x3d_stem_mfma
    monkeypatch.setenv("PASN_NO_FC_MFMA", "1")    # ... and so is the matrix-core first conv the unfused pair would take
    torch.manual_seed(21)
    n, c = 2, 24
    x = torch.randn(n, 3, *thw)
    from protoasnet_amd.backbones import _X3DStem

    stem = _X3DStem(c)
    with torch.no_grad():
        stem.bn.weight.uniform_(0.5, 1.5)
        stem.bn.bias.normal_(0, 0.3)
        stem.bn.running_mean.normal_(0, 0.3)
        stem.bn.running_var.uniform_(0.5, 1.5)
    stem.eval()
    xy = _rt(stem.conv_xy(_rt(x, in_dtype)), dtype)
    ref = F.relu(stem.bn(stem.conv_t(xy))).detach()
    stem = stem.to(DEV)
    xin = x.to(DEV).to(in_dtype).contiguous()

    def run():
        pb = _pb(dtype, in_dtype)
        xa = pb.input(tuple(x.shape))
        y = pb.x3d_stem(xa, stem.conv_xy, stem.conv_t, stem.bn)
        kinds = [m["kind"] for m in pb.meta]
        return _run_single(pb, xa, y, xin), kinds

    fused, kinds = run()
    assert kinds == ["stem"]
    atol, rtol = _tols(dtype)
    assert_close(_from_cl(fused, c), ref, atol * max(1.0, float(ref.abs().max())), rtol, f"fused stem {thw}")
    monkeypatch.setenv("PASN_NO_STEM", "1")
    unfused, kinds = run()
    assert kinds == ["first_conv", "dwconv"]
    assert torch.equal(fused, unfused), "fused stem must be bit-identical to the unfused pair"


@pytest.mark.parametrize("in_dtype", [torch.float32, torch.bfloat16, torch.uint8])
@pytest.mark.parametrize("thw,grey", [((7, 18, 24), False), ((3, 10, 132), False), ((1, 16, 16), False), ((12, 9, 20), True), ((16, 12, 12), False)])
def test_x3d_stem_mfma(thw, grey, in_dtype, monkeypatch):
    """X3D stem on the matrix cores (both convs as one MFMA map, T-marching LDS ring): against torch and against the VALU stem; ragged
    row / column tiles, T below / above the ring length, two column tiles (66 output columns), grey clips with the normalisation at load."""
    if in_dtype == torch.uint8 and not grey:
        pytest.skip("uint8 clips enter through the grey pipeline")
    torch.manual_seed(thw[0] + thw[2])
    n, c, cin = 2, 24, 1 if grey else 3
    if in_dtype == torch.uint8:
        x8 = torch.randint(0, 256, (n, 1, *thw), dtype=torch.uint8)
        xin, mean, std, sc = x8, 0.099, 0.171, 255.0
        xf = (x8.float() / 255.0 - mean) / std
    else:
        xf = _rt(torch.randn(n, cin, *thw), in_dtype)
        xin, mean, std, sc = xf.to(in_dtype), 0.0, 1.0, 1.0
    from protoasnet_amd.backbones import _X3DStem

    stem = _X3DStem(c)
    with torch.no_grad():
        stem.bn.weight.uniform_(0.5, 1.5)
        stem.bn.bias.normal_(0, 0.3)
        stem.bn.running_mean.normal_(0, 0.3)
        stem.bn.running_var.uniform_(0.5, 1.5)
    stem.eval()
    x3 = xf.expand(n, 3, *thw) if grey else xf
    ref = F.relu(stem.bn(stem.conv_t(stem.conv_xy(x3)))).detach()
    stem = stem.to(DEV)

    def run(no_mfma):
        monkeypatch.setenv("PASN_NO_STEM_MFMA", "1" if no_mfma else "0")
        pb = _pb(torch.bfloat16, in_dtype)
        if grey:
            pb.in_affine = (1.0 / (sc * std), -mean / std)
        xa = pb.input((n, cin, *thw))
        y = pb.x3d_stem(xa, stem.conv_xy, stem.conv_t, stem.bn)
        return _run_single(pb, xa, y, xin.to(DEV).contiguous()).clone(), pb.meta[-1]["kernel"]

    out, name = run(False)
    assert name.startswith("x3d_stem_mfma_kernel"), name
    old, old_name = run(True)
    assert old_name.startswith("x3d_stem_kernel"), old_name
    scale = max(1.0, float(ref.abs().max()))
    assert_close(_from_cl(out, c), ref, 3e-2 * scale, 2e-2, f"mfma stem {thw} {in_dtype}")
    assert_close(_from_cl(out, c), _from_cl(old, c), 3e-2 * scale, 2e-2, f"mfma stem vs VALU stem {thw} {in_dtype}")


@pytest.mark.parametrize("layer", ["c133_64_144", "c311_144_64", "first_7x7", "x3d_stem"])
def test_matrix_core_kernels_at_full_benchmark_shapes(layer, monkeypatch):
    """The round-2 matrix-core kernels at the FULL shapes of the benchmark configs (8 x 32 x 56 x 56 stage-1 layers of R(2+1)D-18, its
    7x7 stem on 8 x 3 x 32 x 112 x 112, the X3D stem on 4 x 3 x 16 x 224 x 224), where no CPU oracle finishes in seconds: each must agree
    with the kernel it replaces (same operands, fp32 accumulation in another order: one bf16 ulp of the output scale), be bitwise
    reproducible, and keep the padded channels zero."""
    torch.manual_seed(3)
    dtype = torch.bfloat16

    def bn_of(c):
        bn = nn.BatchNorm3d(c)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.3)
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.5, 1.5)
        return bn.eval().to(DEV)

    if layer in ("c133_64_144", "c311_144_64"):
        cin, cout, k, p = (64, 144, (1, 3, 3), (0, 1, 1)) if layer == "c133_64_144" else (144, 64, (3, 1, 1), (1, 0, 0))
        x = torch.randn(8, cin, 32, 56, 56)
        conv, bn = nn.Conv3d(cin, cout, k, 1, p, bias=False).to(DEV), bn_of(cout)
        env, new_name, old_name = "PASN_NO_HALO", "igemm_halo_kernel", "igemm_glds_kernel"
        if layer == "c311_144_64":  # round 5: the temporal layer's default is the weight-stationary T-marching kernel, held against the halo kernel
            env, new_name, old_name = "PASN_NO_TCONV", "tconv_ws_kernel", "igemm_halo_kernel"

        def run():
            pb = _pb(dtype)
            xa, xs = _cl_input(pb, x, dtype)
            y = pb.conv(xa, conv, bn, "relu")
            return pb.finish(xa, y).run(xs).clone(), pb.meta[0]["kernel"], cout
    elif layer == "first_7x7":
        x = torch.randn(8, 3, 32, 112, 112)
        conv, bn = nn.Conv3d(3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), bias=False).to(DEV), bn_of(45)
        env, new_name, old_name = "PASN_NO_FC_MFMA", "first_conv_mfma_kernel", "first_conv_kernel"

        def run():
            pb = _pb(dtype, dtype)
            xa = pb.input(tuple(x.shape))
            y = pb.first_conv(xa, conv, bn, "relu")
            return _run_single(pb, xa, y, x.to(DEV).to(dtype).contiguous()).clone(), pb.meta[-1]["kernel"], 45
    else:
        from protoasnet_amd.backbones import _X3DStem

        x = torch.randn(4, 3, 16, 224, 224)
        stem = _X3DStem(24)
        stem.bn = bn_of(24)
        stem = stem.eval().to(DEV)
        env, new_name, old_name = "PASN_NO_STEM_MFMA", "x3d_stem_mfma_kernel", "x3d_stem_kernel"

        def run():
            pb = _pb(dtype, dtype)
            xa = pb.input(tuple(x.shape))
            y = pb.x3d_stem(xa, stem.conv_xy, stem.conv_t, stem.bn)
            return _run_single(pb, xa, y, x.to(DEV).to(dtype).contiguous()).clone(), pb.meta[-1]["kernel"], 24

    def switch(off):
        if env == "PASN_NO_TCONV":
            monkeypatch.setenv("PASN_TCONV", "0" if off else "1")
        else:
            monkeypatch.setenv(env, "1" if off else "0")

    switch(False)
    a, name, c = run()
    assert name.startswith(new_name), name
    b, _, _ = run()
    assert torch.equal(a, b), "bitwise reproducible"
    switch(True)
    ref, name_old, _ = run()
    assert name_old.startswith(old_name), name_old
    torch.cuda.synchronize()
    af, rf = a[..., :c].float(), ref[..., :c].float()
    scale = float(rf.abs().max())
    tol = 2.5e-2 if layer in ("first_7x7", "x3d_stem") else 1.6e-2  # the stems multiply bf16-rounded weights (the VALU kernels fp32 ones)
    err = float((af - rf).abs().max())
    assert err <= tol * scale, f"{layer}: max |new - old| = {err:.3g} at output scale {scale:.3g}"
    assert float((af - rf).abs().mean()) <= 2e-3 * scale
    if a.shape[-1] > c:
        assert float(a[..., c:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool(dtype):
    torch.manual_seed(9)
    x = torch.randn(2, 64, 1, 15, 14)
    ref = F.max_pool3d(_rt(x, dtype), (1, 3, 3), (1, 2, 2), (0, 1, 1))
    pb = _pb(dtype)
    xa, xs = _cl_input(pb, x, dtype)
    y = pb.maxpool(xa, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    out = _run_single(pb, xa, y, xs)
    assert_close(_from_cl(out, 64), ref, 0, 0, "maxpool is exact")


def test_bad_arguments_raise():
    """Error behaviour of the boundary: bad geometry -> ValueError with the library's message, no launch."""
    from protoasnet_amd import _lib

    d = _lib.ConvDesc(N=1, Ti=1, Hi=4, Wi=4, Cin=5, Cin_p=5, To=1, Ho=4, Wo=4, Cout=8, Cout_p=8, kt=1, kh=1, kw=1, st=1, sh=1, sw=1)
    t = torch.zeros(1024, device=DEV)
    with pytest.raises(ValueError, match="multiples of 8"):
        _lib.check(_lib.lib().pasn_conv3d_fwd(t.data_ptr(), t.data_ptr(), 0, 0, 0, 0, t.data_ptr(), ctypes.byref(d), 0, 0))
    with pytest.raises(ValueError, match="null pointer"):
        _lib.check(_lib.lib().pasn_maxpool3d_fwd(0, 0, ctypes.byref(d), 0, 0))
