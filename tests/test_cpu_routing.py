"""CPU: the routing is frozen.  (a) the library reads the PASN_* switches from one snapshot of the environment, registered names only
(csrc/tuning.h); (b) every switch the sources consult is in the registry and in DESIGN.md's table; (c) the default launch list of the
benchmarked configuration equals the committed snapshot (tests/golden/routing_x3d_s_cfg2.json)."""
import glob
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import GOLDEN, REPO
from protoasnet_amd import _lib


def _registry():
    rows = [ln.split("\t") for ln in _lib.tuning_report(with_registry=True).splitlines() if ln.count("\t") == 2]
    return {r[0]: (r[1], r[2]) for r in rows}


def test_every_switch_the_sources_read_is_registered_and_documented():
    reg = _registry()
    assert len(reg) >= 60 and all(c in ("route", "geom", "dev") for c, _ in reg.values())
    used = {}
    for path in glob.glob(os.path.join(REPO, "protoasnet_amd", "csrc", "*.hip")) + glob.glob(os.path.join(REPO, "protoasnet_amd", "csrc", "*.h")):
        if os.path.basename(path) in ("tuning.hip", "tuning.h"):
            continue
        text = open(path).read()
        assert not re.search(r'\bgetenv\s*\(\s*"PASN_', text), f"{path}: reads the environment directly; use tune() / tune_dev()"
        for fn, name in re.findall(r'\b(tune|tune_dev|tune_is)\s*\(\s*"(PASN_[A-Z0-9_]+)"', text):
            used.setdefault(name, set()).add(fn)
    for path in glob.glob(os.path.join(REPO, "protoasnet_amd", "*.py")):
        for name in re.findall(r'tuning_get\(\s*"(PASN_[A-Z0-9_]+)"', open(path).read()):
            used.setdefault(name, set()).add("tune")
    for name, fns in used.items():
        assert name in reg, f"{name} is consulted but not in the registry (csrc/tuning.hip)"
        # a dev knob (timing ablation / uncovered geometry) must be compiled out of the product build, and only a dev knob may be
        assert ("tune_dev" in fns) == (reg[name][0] == "dev"), (name, fns, reg[name][0])
    design = open(os.path.join(REPO, "DESIGN.md")).read()
    missing = [n for n, (c, _) in reg.items() if c != "dev" and f"`{n}`" not in design]
    assert not missing, f"switches of the product build missing from DESIGN.md's table: {missing}"


def test_switches_are_read_from_one_snapshot():
    code = (
        "import os\n"
        "os.environ['PASN_DWMFMA'] = '0'; os.environ['PASN_WS_ABL'] = '3'; os.environ['PASN_TYPO_SWITCH'] = '1'\n"
        "from protoasnet_amd import _lib\n"
        "assert _lib.tuning_get('PASN_DWMFMA') == '0'\n"
        "assert _lib.tuning_get('PASN_WS_ABL') is None            # dev knob: compiled out of the product build\n"
        "rep = _lib.tuning_report()\n"
        "assert 'PASN_DWMFMA=0' in rep and 'unknown: PASN_TYPO_SWITCH' in rep and 'PASN_WS_ABL' not in rep, rep\n"
        "os.environ['PASN_DWMFMA'] = '1'; os.environ['PASN_NO_XPAIR'] = '1'\n"
        "assert _lib.tuning_get('PASN_DWMFMA') == '0' and _lib.tuning_get('PASN_NO_XPAIR') is None   # not seen: one snapshot\n"
        "_lib.tuning_reload()\n"
        "assert _lib.tuning_get('PASN_DWMFMA') == '1' and _lib.tuning_get('PASN_NO_XPAIR') == '1'\n"
        "with _lib.tuning_env(PASN_NO_XPAIR=None, PASN_EXPDW='0'):\n"
        "    assert _lib.tuning_get('PASN_NO_XPAIR') is None and _lib.tuning_get('PASN_EXPDW') == '0'\n"
        "assert _lib.tuning_get('PASN_NO_XPAIR') == '1' and _lib.tuning_get('PASN_EXPDW') is None\n"
        "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PASN_")}
    env["PYTHONPATH"] = REPO
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_default_routing_of_the_benchmarked_configuration_is_the_committed_snapshot(monkeypatch):
    """BASELINE config 2 (X3D-S, 32 x 16 x 224 x 224, bf16) compiled with no switch set: launch count, plan kind, kernel instance and
    layer shape of every launch.  A change here changes what bench.py measures: regenerate the snapshot on purpose
    (tests/golden/make_routing_snapshot.py), never by accident."""
    for k in [k for k in os.environ if k.startswith("PASN_")]:
        monkeypatch.delenv(k)
    _lib.tuning_reload()
    sys.path.insert(0, GOLDEN)
    import make_routing_snapshot

    want = json.load(open(os.path.join(GOLDEN, "routing_x3d_s_cfg2.json")))
    got = make_routing_snapshot.routing()
    assert len(got) == want["launches"], f"{len(got)} launches, snapshot has {want['launches']}"
    for i, (g, w) in enumerate(zip(got, want["rows"])):
        assert g == w, f"launch {i}: {g} != snapshot {w}"
    # and a switch really moves it (the comparison is not vacuous)
    monkeypatch.setenv("PASN_EXPDW", "0")
    assert [r["kernel"] for r in make_routing_snapshot.routing()] != [r["kernel"] for r in want["rows"]]
