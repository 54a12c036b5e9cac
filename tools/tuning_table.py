#!/usr/bin/env python3
"""Rewrites the switch table of DESIGN.md (between the TUNING-TABLE markers) from the library's registry (csrc/tuning.hip)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from protoasnet_amd import _lib  # noqa: E402

rows = [ln.split("\t") for ln in _lib.tuning_report(with_registry=True).splitlines() if ln.count("\t") == 2]
table = ["| switch | class | meaning |", "|---|---|---|"] + [f"| `{n}` | {c} | {d} |" for n, c, d in rows]
path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
new = re.sub(r"(<!-- TUNING-TABLE -->\n).*?(<!-- /TUNING-TABLE -->)", lambda m: m.group(1) + "\n".join(table) + "\n" + m.group(2), text, flags=re.S)
assert new != text or "\n".join(table) in text, "markers not found in DESIGN.md"
open(path, "w").write(new)
print(f"{len(rows)} switches")
