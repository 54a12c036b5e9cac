"""CPU oracle for the ProtoASNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker or as
the reported CPU baseline.  ``protoasnet_amd`` never imports it and has no CPU
fallback: without the HIP library it raises.

What it is: a restatement, in plain fp32 ``torch`` ops on the CPU, of the op
sequence the reference runs on its hot path (``/root/reference/src/models`` and
the selection loops of ``src/utils/push_*.py``).  Every function cites the
reference ``file:line`` it follows.  All functions are *functional over a
state_dict* (``name -> tensor``), so the same parameters can be fed to the
reference, to this oracle and to the HIP modules.

Pin status (DESIGN.md section "Oracle"):

* PINNED by golden vectors produced by importing the reference in the build
  container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``):
  ResNet-18 feature trunk, PPNet L2 head (``forward`` / ``push_forward``),
  XProtoNet 2-D head, Video_XProtoNet 3-D head, constructor semantics
  (class identity, last-layer init, receptive-field info), both push
  selection rules (restated; fed with reference ``push_forward`` outputs).
* PARITY UNPINNED: the R(2+1)D-18 trunk (its arithmetic lives in third-party
  ``torchvision.models.video.r2plus1d_18`` -- torchvision 0.14.1 per the
  reference's docker image -- which is not installed here) and the X3D trunk
  (not in the reference at all; BASELINE.json names it).  Both are restated from
  their published definitions and checked for self-consistency only.
"""

from . import backbones, heads, losses, nets, push, receptive_field, trainer  # noqa: F401
