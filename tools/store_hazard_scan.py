#!/usr/bin/env python3
"""Scan gfx950 assembly for a wide buffer store whose data registers are overwritten in the next instructions.

A `buffer_store_dwordx3/x4` holds its data VGPRs for some cycles after issue.  LLVM's hazard recognizer inserts the wait states when the
store's soffset is an immediate, and NOT when it is an SGPR (GCNHazardRecognizer::createsVALUHazard) -- on gfx950 the hazard exists with an
SGPR soffset too (round 5: dw_tz.hip's first store, data v[46:49], followed by `v_cndmask_b32 v46`: lanes of the stored rows took the new value
in a few percent of the runs).  Usage: hipcc -S --cuda-device-only ... -o k.s ; python tools/store_hazard_scan.py k.s [...]
"""
import re
import sys

STORE = re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(v\d+|off),\s*s\[\d+:\d+\],\s*(s\d+|\d+|0x[0-9a-f]+)")
WRITE = re.compile(r"^\s*(v_\w+|ds_read\w*|buffer_load\w*|global_load\w*)\s+(v\[(\d+):(\d+)\]|v(\d+))")


def scan(path, window=2):
    lines = [l for l in open(path) if l.strip() and not l.lstrip().startswith((";", ".", "//")) and not l.rstrip().endswith(":")]
    hits = 0
    for i, l in enumerate(lines):
        m = STORE.match(l)
        if not m or not m.group(4).startswith("s"):
            continue
        lo, hi = int(m.group(1)), int(m.group(2))
        for j in range(1, window + 1):
            if i + j >= len(lines):
                break
            n = lines[i + j]
            if n.lstrip().startswith("s_nop"):
                break
            w = WRITE.match(n)
            if w:
                a = int(w.group(3)) if w.group(3) else int(w.group(5))
                b = int(w.group(4)) if w.group(4) else a
                if a <= hi and b >= lo:
                    hits += 1
                    print(f"{path}: {l.strip()}  ->  +{j}: {n.strip()}")
    return hits


if __name__ == "__main__":
    sys.exit(1 if sum(scan(p) for p in sys.argv[1:]) else 0)
