"""CPU: structural pins of the two trunks whose arithmetic lives outside /root/reference (torchvision's r2plus1d_18, the X3D
recipe of pytorchvideo): neither library is installed here and there is no network, so their graphs cannot be run.  What CAN
be pinned without them are the published totals their definitions imply -- parameter counts, checkpoint key names and tensor
shapes, multiply-accumulate totals -- which fail on any wrong width, kernel size, stride, bias or block count."""
import re

import torch

from protoasnet_amd import backbones, plan


def _plan_totals(trunk, shape, dtype=torch.bfloat16):
    pb = plan.PlanBuilder(torch.device("cpu"), dtype, dtype)  # geometry only: nothing is launched
    x = pb.input(shape)
    with torch.no_grad():
        y = trunk.build_plan(pb, x)
    macs = sum(m["flops"] for m in pb.meta) / 2 / shape[0]
    return macs, (y.C, y.T, y.H, y.W), len(pb.ops)


def test_r2plus1d_18_full_graph_has_torchvisions_parameter_count_and_checkpoint_layout():
    """torchvision.models.video.r2plus1d_18 (0.14.1, the reference's docker image): 31 505 325 parameters (its documented
    ``num_params``), children stem / layer1-4 / avgpool / fc; the reference wraps ``children()[:last_layer_num]`` in a Sequential
    named ``backbone`` (resnet_features.py:316-320), so checkpoint key ``layerN.x`` appears as ``backbone.N.x``, ``stem.x`` as ``backbone.0.x``."""
    full = backbones.resnet2p1d_18(pretrained=False, last_layer_num=-2)  # stem + layer1..4
    n = sum(p.numel() for p in full.parameters())
    assert n + 512 * 400 + 400 == 31_505_325  # + fc(512 -> 400 Kinetics classes)
    sd = {re.sub(r"^backbone\.(\d)", lambda m: "stem" if m.group(1) == "0" else f"layer{m.group(1)}", k): tuple(v.shape)
          for k, v in full.state_dict().items()}
    want = {  # torchvision VideoResNet / R2Plus1dStem / Conv2Plus1D / BasicBlock: midplanes = (in*out*27)//(in*9+3*out)
        "stem.0.weight": (45, 3, 1, 7, 7), "stem.1.running_mean": (45,), "stem.3.weight": (64, 45, 3, 1, 1), "stem.4.weight": (64,),
        "layer1.0.conv1.0.0.weight": (144, 64, 1, 3, 3), "layer1.0.conv1.0.1.weight": (144,), "layer1.0.conv1.0.3.weight": (64, 144, 3, 1, 1),
        "layer1.0.conv1.1.weight": (64,), "layer1.1.conv2.0.3.weight": (64, 144, 3, 1, 1), "layer1.1.conv2.1.bias": (64,),
        "layer2.0.conv1.0.0.weight": (230, 64, 1, 3, 3), "layer2.0.conv1.0.3.weight": (128, 230, 3, 1, 1),
        "layer2.0.conv2.0.0.weight": (230, 128, 1, 3, 3), "layer2.0.downsample.0.weight": (128, 64, 1, 1, 1),
        "layer2.0.downsample.1.running_var": (128,), "layer2.1.conv1.0.0.weight": (288, 128, 1, 3, 3),
        "layer3.0.conv1.0.0.weight": (460, 128, 1, 3, 3), "layer3.0.conv1.0.3.weight": (256, 460, 3, 1, 1),
        "layer3.1.conv2.0.0.weight": (576, 256, 1, 3, 3), "layer3.0.downsample.0.weight": (256, 128, 1, 1, 1),
        "layer4.0.conv1.0.0.weight": (921, 256, 1, 3, 3), "layer4.1.conv1.0.0.weight": (1152, 512, 1, 3, 3),
        "layer4.1.conv2.1.num_batches_tracked": (),
    }
    for k, shp in want.items():
        assert sd.get(k) == shp, (k, sd.get(k), shp)
    assert not any("bias" in k and re.search(r"\.(0|3)\.bias$", k) and "conv" in k for k in sd)  # convs carry no bias
    assert len(full.state_dict()) == 222 and "layer1.0.downsample.0.weight" not in sd
    # the reference's own slice: [:-3] = stem + layer1-3, 256 channels (resnet_features.py:311-313)
    ref_slice = backbones.resnet2p1d_18(pretrained=False, last_layer_num=-3)
    assert ref_slice.out_channels == 256 and len(ref_slice.backbone) == 4
    assert sum(p.numel() for p in ref_slice.parameters()) == 7_804_601


def test_r2plus1d_18_multiply_accumulate_totals():
    """76.02 GMAC per 32x112x112 clip, 152.04 per 16x224x224 clip, output (256, T/4, H/8, W/8)."""
    trunk = backbones.resnet2p1d_18(pretrained=False)
    macs, out, _ = _plan_totals(trunk, (1, 3, 32, 112, 112))
    assert out == (256, 8, 14, 14) and abs(macs / 1e9 - 76.02) < 0.005
    macs, out, _ = _plan_totals(trunk, (1, 3, 16, 224, 224))
    assert out == (256, 4, 28, 28) and abs(macs / 1e9 - 152.04) < 0.005


def test_x3d_graph_has_the_published_size():
    """pytorchvideo's x3d_s / x3d_m (one network, two input sizes): 3.79 M parameters for the 400-class model = trunk up to res5
    (this package) + conv5 192->432 + BN + the 432->2048 head conv + the 2048->400 projection."""
    for arch in ("x3d_s", "x3d_m"):
        trunk = backbones.X3DFeatures(arch)
        n = sum(p.numel() for p in trunk.parameters())
        head = 192 * 432 + 2 * 432 + 432 * 2048 + 2048 * 400 + 400
        assert n == 2_006_178 and round((n + head) / 1e6, 2) == 3.79
        widths = [[blk.conv_b.in_channels for blk in stage] for stage in trunk.stages]
        assert [len(w) for w in widths] == [3, 5, 11, 7] and [w[0] for w in widths] == [54, 108, 216, 432]
        se = [[blk.se.fc1.out_channels if blk.se is not None else 0 for blk in stage] for stage in trunk.stages]
        assert [s[0] for s in se] == [8, 8, 16, 32] and all(s[1] == 0 for s in se)  # SE on even blocks, ratio 1/16 rounded to 8
        assert trunk.stem.conv_xy.weight.shape == (24, 3, 1, 3, 3) and trunk.stem.conv_t.weight.shape == (24, 1, 5, 1, 1)
        assert all(stage[0].conv_b.stride == (1, 2, 2) and stage[1].conv_b.stride == (1, 1, 1) for stage in trunk.stages)


def test_x3d_multiply_accumulate_totals():
    """4.66 GMAC per 16x224x224 clip (X3D-S), ~18.6 per 32x312x312 clip (X3D-M); features (192, T, H/32, W/32)."""
    macs, out, _ = _plan_totals(backbones.X3DFeatures("x3d_s"), (2, 3, 16, 224, 224))
    assert out == (192, 16, 7, 7) and abs(macs / 1e9 - 4.666) < 0.005
    macs, out, _ = _plan_totals(backbones.X3DFeatures("x3d_m"), (1, 3, 32, 312, 312))
    assert out == (192, 32, 10, 10) and abs(macs / 1e9 - 18.63) < 0.01


def test_resnet18_multiply_accumulate_total():
    macs, out, _ = _plan_totals(backbones.resnet18_features(pretrained=False), (1, 3, 224, 224))
    assert out == (512, 1, 7, 7) and abs(macs / 1e9 - 1.8136) < 0.001


def test_oracle_trunks_consume_exactly_these_graphs():
    """The oracle's functional restatements read every key of the full graphs (nothing ignored, nothing extra) and produce the
    published output geometry: r2plus1d_18 children[:-2] -> (512, T/8, H/16, W/16), X3D res5 -> (192, T, H/32, W/32)."""
    import oracle
    from protoasnet_amd import synth

    class Spy(dict):
        def __init__(self, d):
            super().__init__(d)
            self.read = set()

        def __getitem__(self, k):
            self.read.add(k)
            return super().__getitem__(k)

    full = synth.load_synth(backbones.resnet2p1d_18(pretrained=False, last_layer_num=-2)).eval()
    sd = Spy({"cnn_backbone." + k: v for k, v in full.state_dict().items()})
    with torch.no_grad():
        y = oracle.backbones.r2plus1d_18_trunk(sd, "cnn_backbone.", synth.echo_clips((1, 3, 8, 32, 32)), last_layer_num=-2)
    assert tuple(y.shape) == (1, 512, 1, 2, 2)
    unread = {k for k in sd if k not in sd.read and not k.endswith("num_batches_tracked")}
    assert not unread, sorted(unread)[:5]
    x3d = synth.load_synth(backbones.X3DFeatures("x3d_s")).eval()
    sd = Spy({"cnn_backbone." + k: v for k, v in x3d.state_dict().items()})
    with torch.no_grad():
        y = oracle.backbones.x3d_trunk(sd, "cnn_backbone.", synth.echo_clips((1, 3, 4, 64, 64)))
    assert tuple(y.shape) == (1, 192, 4, 2, 2)
    unread = {k for k in sd if k not in sd.read and not k.endswith("num_batches_tracked")}
    assert not unread, sorted(unread)[:5]
