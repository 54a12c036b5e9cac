"""Device side of the input pipeline.

The reference's dataset does, per clip and on the host (``src/data/as_dataloader.py:184-231``): load the cine, resize to
(T, H, W) in [0, 1] (skimage), optional augmentation, ``bin_to_norm`` ((x - 0.099) / 0.171, :173-182), ``gray_to_gray3``
(expand to 3 identical channels, :168-170), ``.float()``.  The last three steps triple the bytes that cross PCIe and that the first
conv reads.  Here the clip stays SINGLE-channel (fp32, bf16 or uint8) until it is on the GPU: normalisation is fused into the first
layer's loads (``HipTrunk.set_input_normalization``) and the channel expansion is folded into the first conv's weights (summed over
the input channels: 9 spatial taps instead of 27 for the X3D stem) -- ``pasn_x3d_stem_gray_fwd`` / ``pasn_first_conv_gray_fwd``.

``bin_to_norm`` / ``gray_to_gray3`` keep the reference's names and semantics for code that still wants the 3-channel tensor.
"""
from __future__ import annotations

from typing import Optional

import torch

ECHO_MEAN, ECHO_STD = 0.099, 0.171  # as_dataloader.py:180-181


def gray_to_gray3(in_tensor: torch.Tensor) -> torch.Tensor:
    """1xTxHxW -> 3xTxHxW view (as_dataloader.py:166-170)."""
    return in_tensor.expand(3, *([-1] * (in_tensor.dim() - 1)))


def bin_to_norm(in_tensor: torch.Tensor) -> torch.Tensor:
    """[0, 1] pixels -> normalised (as_dataloader.py:172-182)."""
    return (in_tensor - ECHO_MEAN) / ECHO_STD


class DeviceClipPipeline:
    """Batches of single-channel clips -> what ``model(x)`` takes, with the normalisation and channel expansion left to the GPU.

    ``pipe = DeviceClipPipeline(model, normalize=True)`` configures the model's trunk once; ``x = pipe(cine)`` takes a host or
    device batch shaped (N,T,H,W) / (N,1,T,H,W) (video) or (N,H,W) / (N,1,H,W) (image) in [0, 1] (float) or [0, 255] (uint8), moves
    it to the model's device asynchronously (one third of the reference's bytes, a twelfth for uint8) and returns the (N,1,...)
    tensor the HIP trunk accepts directly.  ``normalize=False``: the clip is already normalised.
    """

    def __init__(self, model: torch.nn.Module, normalize: bool = True, video: Optional[bool] = None):
        self.trunk = getattr(model, "cnn_backbone", None) or getattr(model, "features")
        self.device = next(model.parameters()).device
        self.normalize = normalize
        self.video = type(model).__name__.startswith("Video") if video is None else bool(video)

    def __call__(self, cine: torch.Tensor) -> torch.Tensor:
        x = cine
        if x.dim() == (4 if self.video else 3):  # (N,T,H,W) / (N,H,W): add the channel axis
            x = x.unsqueeze(1)
        if x.dim() != (5 if self.video else 4) or x.shape[1] != 1:
            raise ValueError("DeviceClipPipeline takes single-channel clips (N,[1,]%sH,W); a 3-channel tensor goes to the model as it is"
                             % ("T," if self.video else ""))
        if x.dtype not in (torch.uint8, torch.float32, torch.bfloat16):
            x = x.float()
        scale = 1.0 / 255.0 if x.dtype == torch.uint8 else 1.0
        if self.normalize:
            self.trunk.set_input_normalization(ECHO_MEAN, ECHO_STD, scale)
        else:
            self.trunk.set_input_normalization(None)
        return x.contiguous().to(self.device, non_blocking=True)
