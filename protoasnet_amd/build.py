"""Build ``libprotoasnet_amd.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
SOURCES = ("api.hip", "tuning.hip", "comm.hip", "conv.hip", "pwconv.hip", "gemm_pw.hip", "igemm.hip", "igemm_halo.hip", "first_conv_mfma.hip", "pwconv_xtile.hip", "pwconv_tiny.hip", "pwconv_ws.hip", "pwconv_xpair.hip", "tconv_ws.hip", "stem.hip", "stem_mfma.hip", "dwmarch.hip", "dwmfma.hip", "x3d_expdw.hip", "x3d_expdw_tz.hip", "dw_tz.hip", "dwtemporal.hip", "x3d_pe.hip", "x3d_edp.hip", "head_l2.hip", "head_xproto.hip", "head_chain.hip", "push.hip", "train.hip", "pack.hip", "wgrad.hip", "wgrad_halo.hip", "head_train.hip", "warp.hip")
ARCH = "gfx950"


def lib_path(variant: str = "") -> str:
    return os.path.join(OUT_DIR, f"libprotoasnet_amd{'_' + variant if variant else ''}.so")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force: bool = False, verbose: bool = False, variant: str = "") -> str:
    """Compile every HIP source for gfx950 and link the shared library; returns its path.

    ``variant="tuning"``: a second library, ``libprotoasnet_amd_tuning.so`` (own object directory), built with ``-DPASN_TUNING
    -DPASN_WS_ABLATE``: the dev-class switches of csrc/tuning.h (timing ablations, unswept geometry) are compiled in.  Never the product:
    select it per process with ``PASN_LIB_PATH`` (tools/*.sh)."""
    hipcc = _hipcc()
    obj_dir = OBJ_DIR + ("_" + variant if variant else "")
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    # every header of csrc/ (igemm_epilogue.h is shared by three kernels' files: an edit there must rebuild them) + the C-ABI header
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(os.path.dirname(HERE), "include", "protoasnet_amd.h")]
    flags_extra = os.environ.get("PASN_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DPASN_WS_ABLATE for tools/ws_abl.sh (use with force=True / a clean obj dir)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + flags_extra
    if variant == "tuning":
        flags += ["-DPASN_TUNING", "-DPASN_WS_ABLATE"]
    elif variant:
        raise ValueError(f"unknown build variant {variant!r}")

    def compile_one(src: str) -> str:
        s, o = os.path.join(CSRC, src), os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return o

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    target = lib_path(variant)
    if force or _stale(target, objs):
        r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", target] + objs + ["-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return target
