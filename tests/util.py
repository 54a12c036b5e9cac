"""Shared helpers for the tests: state_dicts from the synthetic recipe, golden-style inputs."""
import numpy as np
import torch

from protoasnet_amd import synth


def head_b_state(cb, d, p, k, video=True):
    """state_dict of the reference's head B (Video_XProtoNet.py:27-80 / XProtoNet.py:17-46) from names+shapes."""
    one = (1, 1, 1) if video else (1, 1)
    shapes = {
        "add_on_layers.0.weight": (d, cb) + one, "add_on_layers.0.bias": (d,),
        "add_on_layers.2.weight": (d, d) + one, "add_on_layers.2.bias": (d,),
        "occurrence_module.0.weight": (d, cb) + one, "occurrence_module.0.bias": (d,),
        "occurrence_module.2.weight": (d // 2, d) + one, "occurrence_module.2.bias": (d // 2,),
        "occurrence_module.4.weight": (p, d // 2) + one,
        "prototype_vectors": (p, d) + one, "ones": (p, d) + one, "last_layer.weight": (k, p),
    }
    return {n: torch.from_numpy(synth.synth_tensor(n, s)) for n, s in shapes.items()}


def video_features(shape, seed):
    """Same recipe as tests/golden/make_golden.py::video_features."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal(tuple(shape)).astype(np.float32)
    return torch.from_numpy(np.maximum(f, 0.0))


def synth_model(cfg):
    from protoasnet_amd import model_builder

    m = model_builder.build(cfg)
    synth.load_synth(m)
    return m.eval()


CFG_PPNET = dict(checkpoint_path="", name="ProtoPNet", base_architecture="resnet18", pretrained=False,
                 prototype_shape="(30, 512, 1, 1)", num_classes=3, img_size=224, add_on_layers_type="regular",
                 prototype_activation_function="log")
CFG_PPNET_BOTTLENECK = dict(CFG_PPNET, prototype_shape="(12, 128, 1, 1)", add_on_layers_type="bottleneck")
CFG_XPROTO = dict(checkpoint_path="", name="XProtoNet", base_architecture="resnet18", pretrained=False,
                  prototype_shape="(40, 512, 1, 1)", num_classes=4, img_size=224, add_on_layers_type="regular")
CFG_VIDEO_R2P1D = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture="resnet2p1d_18", backbone_last_layer_num=-3,
                       pretrained=False, prototype_shape="(40, 256, 1, 1, 1)", num_classes=4, img_size=112)
CFG_VIDEO_X3D = dict(checkpoint_path="", name="Video_XProtoNet", base_architecture="x3d_s", backbone_last_layer_num=-3,
                     pretrained=False, prototype_shape="(30, 256, 1, 1, 1)", num_classes=3, img_size=224)


# ---- push loaders of the G4 fixtures: the table in tests/golden/make_golden_push.py::push_recipe, restated as data --------
PUSH_RECIPE = {
    "image": ([[10, 11, 12, 13], [20, 21, 22, 23], [30, 31, 32, 30], [40, 41, 42, 43], [20, 21, 22, 23], [50, 51, 52, 53]],
              [[0, 1, 2, 0], [1, 2, 0, 1], [2, 2, 1, 2], [1, 1, 1, 1], [1, 2, 0, 1], [0, 0, 2, 1]]),
    "video": ([[110, 111, 112], [120, 121, 122], [130, 131, 130], [120, 121, 122], [140, 141, 142]],
              [[0, 1, 2], [2, 0, 1], [1, 1, 1], [2, 0, 1], [0, 2, 2]]),
}


class PushLoader:
    """len / iteration over dict samples / ``batch_size``: all the push routines need of a DataLoader."""

    def __init__(self, batches, batch_size):
        self.batches, self.batch_size = batches, batch_size

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def push_loader(kind, shape):
    seeds, labels = PUSH_RECIPE[kind]
    out = []
    for bi, (ss, ls) in enumerate(zip(seeds, labels)):
        if kind == "video":
            x = torch.stack([video_features(shape, s) for s in ss])
        else:
            x = torch.cat([synth.echo_clips((1,) + tuple(shape), seed=s) for s in ss])
        out.append({"cine": x, "target_AS": torch.tensor(ls, dtype=torch.int64), "filename": [f"b{bi}_{a}" for a in range(len(ss))]})
    return PushLoader(out, len(seeds[0]))


CFG_PUSH_XIMG = dict(checkpoint_path="", name="XProtoNet", base_architecture="resnet18", pretrained=False,
                     prototype_shape="(12, 32, 1, 1)", num_classes=4, img_size=64, add_on_layers_type="regular")
CFG_PUSH_PPNET = dict(checkpoint_path="", name="ProtoPNet", base_architecture="resnet18", pretrained=False,
                      prototype_shape="(6, 32, 1, 1)", num_classes=3, img_size=64, add_on_layers_type="regular",
                      prototype_activation_function="log")
