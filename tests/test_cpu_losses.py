"""CPU: the oracle's restatement of torchvision's affine warp (TransformLoss) against properties that hold for any correct
implementation -- torchvision itself is absent from this image, so there are no reference outputs to pin it with (oracle/losses.py)."""
import torch

import oracle


def test_affine_identity_and_half_turn():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 9, 12, generator=g)
    assert torch.allclose(oracle.losses.affine(x, 0.0, 1.0), x, atol=1e-6)
    # 180 degrees about the centre maps pixel centres onto pixel centres: an exact double flip
    assert torch.allclose(oracle.losses.affine(x, 180.0, 1.0), x.flip(2, 3), atol=1e-5)


def test_affine_scale_two_is_bilinear_magnification_with_zero_fill():
    x = torch.ones(1, 1, 8, 8)
    y = oracle.losses.affine(x, 0.0, 0.5)  # shrink: the image covers the central 4x4, zeros (fill) around it
    assert float(y[0, 0, 0, 0]) == 0.0 and abs(float(y[0, 0, 4, 4]) - 1.0) < 1e-6
    assert 0.0 < float(y.sum()) < 64.0


def test_transform_loss_is_zero_for_an_equivariant_map():
    """If the "model" is the identity on single-channel images, warp(occ(x)) == occ(warp(x)) and the loss vanishes."""
    x = torch.rand(2, 1, 16, 16)
    occ = x.unsqueeze(2)  # (N, P=1, 1, H, W)
    loss = oracle.losses.transform_loss(x, occ, lambda xt: xt.unsqueeze(2), 17.0, 1.2, loss_weight=1.0)
    assert float(loss) < 1e-4
