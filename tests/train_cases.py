"""Inputs shared by tests/golden/make_golden_train.py (runs the reference in train mode) and the tests that replay it."""
import torch

SHAPE = (3, 3, 96, 96)
FULL_GRADS = ("last_layer.weight", "prototype_vectors", "occurrence_module.4.weight", "add_on_layers.2.bias")


def kink_sparse_(model):
    """+2.5 on the bias of every norm layer that feeds a ReLU (all of ResNet-18's but the downsample ones): keeps pre-activations off
    the ReLU kink so that two fp32 implementations agree on the masks (tests/test_gpu_train.py explains why that matters)."""
    with torch.no_grad():
        for name, mod in model.named_modules():
            if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)) and "downsample" not in name:
                mod.bias += 2.5
    return model


def loss_weights(n, p, k, spatial, seed=5):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, k, generator=g), torch.randn(n, p, generator=g), torch.randn((n, p, 1) + tuple(spatial), generator=g) * 0.1
