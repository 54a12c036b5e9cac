"""``build(model_config)`` -- the factory the reference's agents call (src/models/model_builder.py:7-25).

Same contract: ``name`` selects the constructor, ``checkpoint_path`` is dropped (the agent loads it),
``prototype_shape`` arrives as a *string* from the YAML and is parsed here.  The reference ``eval()``s that
string; this build parses it as a literal tuple instead (same accepted inputs, no code execution).
"""
from __future__ import annotations

import ast
import logging
from copy import deepcopy

from .nets import construct_PPNet, construct_Video_XProtoNet, construct_XProtoNet

MODELS = {
    "ProtoPNet": construct_PPNet,
    "XProtoNet": construct_XProtoNet,
    "Video_XProtoNet": construct_Video_XProtoNet,
}


def build(model_config):
    config = deepcopy(model_config)
    config.pop("checkpoint_path")  # KeyError when absent, like the reference
    shape = config.get("prototype_shape")
    if isinstance(shape, str):
        config["prototype_shape"] = tuple(int(v) for v in ast.literal_eval(shape))
    name = config.pop("name")
    model = MODELS[name](**config)
    logging.info(f"Model {name} is created.")
    return model
