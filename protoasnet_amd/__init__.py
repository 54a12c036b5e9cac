"""protoasnet_amd -- MI355X (gfx950) native hot path of ProtoASNet behind the reference's nn.Module surface.

    from protoasnet_amd import model_builder
    model = model_builder.build(cfg["model"]).cuda().eval()      # same config dict as the reference
    logits, similarity, occurrence_map = model(clips)              # fused HIP kernels underneath

See DESIGN.md for the path, the boundary and the kernels; INTEGRATION.md for the reference-side binding.
"""
from . import synth  # noqa: F401
from .build import build_extension, lib_path  # noqa: F401

__all__ = ["model_builder", "nets", "backbones", "push", "synth", "build_extension", "lib_path"]


def __getattr__(name):  # lazy: importing the package must not need the built library
    if name in ("model_builder", "nets", "backbones", "push", "plan", "receptive_field", "_lib"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
