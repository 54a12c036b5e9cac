"""CPU: the reference's shipped configs drop in unchanged (north_star: "src/configs drop in unchanged").

Every ``src/configs/*.yml`` sets ``pretrained: True`` (``:14-15``); the reference then reads ``./pretrained_models/<file>`` through
``model_zoo.load_url`` (resnet_features.py:8-18,243-247,314-319).  These tests put SYNTHETIC files with the published key layout there
(nothing is copied from the reference; the YAML is read where it lies and skipped when the directory is absent, as on the GPU box).
"""
import glob
import os
import re

import pytest
import torch
import yaml

from protoasnet_amd import backbones, model_builder, synth

REF_CONFIGS = "/root/reference/src/configs"


def _tv_r2plus1d_state():
    """A state_dict with torchvision ``r2plus1d_18``'s key layout: stem.*, layer1-4.*, fc.* (values: the synthetic recipe)."""
    full = synth.load_synth(backbones.resnet2p1d_18(pretrained=False, last_layer_num=-2))  # stem + layer1..4
    sd = {}
    for k, v in full.state_dict().items():
        _, idx, rest = k.split(".", 2)
        sd[("stem." if idx == "0" else f"layer{idx}.") + rest] = v.clone()
    sd["fc.weight"], sd["fc.bias"] = torch.zeros(400, 512), torch.zeros(400)
    return sd


def _tv_resnet18_state():
    sd = {k: v.clone() for k, v in synth.load_synth(backbones.resnet18_features(pretrained=False)).state_dict().items()}
    sd["fc.weight"], sd["fc.bias"] = torch.zeros(1000, 512), torch.zeros(1000)
    return sd


def _model_section(path):
    """(``model:`` mapping, ``data.img_size``) of a reference config.  Baseline_ProtoPNet.yml does not parse as a whole (a stray comma in
    its ``data:`` section, ``:68``), so the top-level ``model:`` block is cut out as text and parsed on its own."""
    text = open(path).read()
    m = re.search(r"^model:.*?\n(?=^\S)", text, flags=re.S | re.M)
    model_config = yaml.safe_load(m.group(0))["model"]
    img_size = int(re.search(r"^data:.*?^  img_size:\s*(\d+)", text, flags=re.S | re.M).group(1))
    return model_config, img_size


@pytest.fixture
def pretrained_dir(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)  # the reference's model_dir is relative to the working directory
    d = tmp_path / "pretrained_models"
    d.mkdir()
    torch.save(_tv_resnet18_state(), str(d / "resnet18-5c106cde.pth"))
    torch.save(_tv_r2plus1d_state(), str(d / "r2plus1d_18-91a641e6.pth"))
    return d


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference configs not present on this box")
def test_every_reference_config_builds_unchanged(pretrained_dir):
    paths = sorted(glob.glob(os.path.join(REF_CONFIGS, "*.yml")))
    assert len(paths) == 6
    for path in paths:
        model_config, img_size = _model_section(path)
        assert model_config["pretrained"] is True
        model_config.update({"img_size": img_size})  # base.py:37-42
        model = model_builder.build(model_config)
        # class names of the reference: ProtoPNet.py:56 ``class PPNet``, XProtoNet.py:14, Video_XProtoNet.py:24
        assert type(model).__name__ == {"ProtoPNet": "PPNet"}.get(model_config["name"], model_config["name"])
        trunk = model.features if hasattr(model, "features") else model.cnn_backbone
        if model_config["base_architecture"] == "resnet18":
            want = _tv_resnet18_state()
            assert torch.equal(trunk.layer3[1].conv2.weight, want["layer3.1.conv2.weight"])
            assert torch.equal(trunk.bn1.running_var, want["bn1.running_var"])
        else:
            want = _tv_r2plus1d_state()
            assert torch.equal(trunk.backbone[0][0].weight, want["stem.0.weight"])
            assert torch.equal(trunk.backbone[3][1].conv2[0][3].weight, want["layer3.1.conv2.0.3.weight"])
            assert len(trunk.backbone) == 7 + model_config["backbone_last_layer_num"]
        # the string prototype_shape of the YAML reached the model as a tuple
        assert model.prototype_shape == tuple(int(v) for v in model_config["prototype_shape"].strip("()").split(","))


def test_pretrained_semantics(pretrained_dir):
    # strict=False: extra keys (layer4 beyond the cut, fc.*) are ignored; every kept tensor comes from the file
    want = _tv_r2plus1d_state()
    trunk = backbones.resnet2p1d_18(pretrained=True, last_layer_num=-3)
    sd = trunk.state_dict()
    assert len(sd) > 100 and not any(k.startswith("backbone.4") for k in sd)
    for k, v in sd.items():
        _, idx, rest = k.split(".", 2)
        assert torch.equal(v, want[("stem." if idx == "0" else f"layer{idx}.") + rest]), k
    r = backbones.resnet18_features(pretrained=True)
    w18 = _tv_resnet18_state()
    for k, v in r.state_dict().items():
        assert torch.equal(v, w18[k]), k
    # the default of the reference's video trunk is pretrained=True (resnet_features.py:308)
    assert torch.equal(backbones.resnet2p1d_18().backbone[0][0].weight, want["stem.0.weight"])
    # fc.* are popped, not ignored: a file without them is a KeyError in the reference too (resnet_features.py:245-246)
    bad = _tv_resnet18_state()
    del bad["fc.bias"]
    torch.save(bad, str(pretrained_dir / "resnet18-5c106cde.pth"))
    with pytest.raises(KeyError):
        backbones.resnet18_features(pretrained=True)
    # a tensor of the wrong shape is an error under strict=False as well
    bad = _tv_r2plus1d_state()
    bad["stem.0.weight"] = torch.zeros(45, 3, 1, 5, 5)
    torch.save(bad, str(pretrained_dir / "r2plus1d_18-91a641e6.pth"))
    with pytest.raises(RuntimeError, match="size mismatch"):
        backbones.resnet2p1d_18(pretrained=True)


def test_missing_pretrained_file_is_named(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError, match=r"pretrained_models/resnet18-5c106cde\.pth"):
        backbones.resnet18_features(pretrained=True)
    with pytest.raises(FileNotFoundError, match=r"pretrained_models/r2plus1d_18-91a641e6\.pth"):
        backbones.resnet2p1d_18(pretrained=True)
    with pytest.raises(RuntimeError, match="pretrained=False"):
        backbones.x3d_s(pretrained=True)
    assert backbones.x3d_s(pretrained=False).out_channels == 192
