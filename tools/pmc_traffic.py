#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes into per-kernel-instance HBM traffic (bytes per launch) for bench.py's `roofline.traffic`.

    rocprofv3 --kernel-trace --mangled-kernels --pmc FETCH_SIZE --output-format csv -d OUT/fetch -o p -- python3 bench.py ...
    rocprofv3 --kernel-trace --mangled-kernels --pmc WRITE_SIZE --output-format csv -d OUT/write -o p -- python3 bench.py ...
    python tools/pmc_traffic.py OUT/fetch/p_counter_collection.csv OUT/write/p_counter_collection.csv [TAG] > profiles/pmc_traffic.json

TAG (e.g. r05c) is stored under "_run" so that bench.py's `roofline.traffic_source` can say which stored run the figure is from.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly HALF of the bytes of wide (16 B/lane) coalesced reads, which is the access shape of
every kernel here, so reads are doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import collections
import csv
import json
import re
import sys


def pretty(mangled: str) -> str:
    """_ZN4pasn21dwconv3d_strip_kernelIDF16bLi7ELi3ELi1EEEv... -> dwconv3d_strip_kernel<bf16,7,3,1>; the profiler prints some names demangled
    ("void pasn::x3d_expdw_kernel<2, 0, 2, false>(...)" -> x3d_expdw_kernel<2,0,2,false>)"""
    dm = re.match(r"(?:void )?pasn::([A-Za-z0-9_]+)(?:<(.*?)>)?\(", mangled)
    if dm:
        args = (dm.group(2) or "").replace(" ", "").replace("__hip_bfloat16", "bf16").replace("float", "f32")
        return f"{dm.group(1)}<{args}>" if args else dm.group(1)
    m = re.match(r"_ZN4pasn\d+([a-z0-9_]+?)I(.*?)EEv", mangled)
    if not m:
        m2 = re.match(r"_ZN4pasn\d+([a-z0-9_]+)", mangled)
        return m2.group(1) if m2 else mangled
    name, args = m.group(1), m.group(2)
    out = []
    for tok in re.finditer(r"DF16b|Li(\d+)E|Lb([01])E|f", args):
        if tok.group(0) == "DF16b":
            out.append("bf16")
        elif tok.group(0) == "f":
            out.append("f32")
        elif tok.group(1) is not None:
            out.append(tok.group(1))
        else:
            out.append("true" if tok.group(2) == "1" else "false")
    return f"{name}<{','.join(out)}>"


def load(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = pretty(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, nw = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if "kernel" not in k:
            continue
        rd = 2.0 * 1024.0 * fetch.get(k, 0.0) / max(nf.get(k, 1), 1)   # KiB -> B, x2 gfx950 wide-read correction
        wr = 1024.0 * write.get(k, 0.0) / max(nw.get(k, 1), 1)
        out[k] = {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "traffic_bytes_per_launch": round(rd + wr), "launches_sampled": int(nf.get(k, 0))}
    if len(sys.argv) > 3:
        out["_run"] = {"tag": sys.argv[3]}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
