"""ctypes binding of ``libprotoasnet_amd.so`` (C-ABI in ``include/protoasnet_amd.h``).

There is deliberately no fallback: if the HIP library is missing or a call fails,
the caller gets an exception.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_long, c_size_t, c_void_p

F32, BF16, U8 = 0, 1, 2
ACT = {"none": 0, "relu": 1, "sigmoid": 2, "swish": 3, "abs": 4}

# (PASN_LIB_PATH: another build of the same C-ABI library -- A/B runs of two commits on one GPU box, tools/ab_bench.sh)
LIB_PATH = os.environ.get("PASN_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libprotoasnet_amd.so")


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "N", "Ti", "Hi", "Wi", "Cin", "Cin_p", "To", "Ho", "Wo", "Cout", "Cout_p",
        "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw", "act", "in_swish", "w_kc", "w_rows", "w_frag",
    )]


class XProtoDesc(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("N", "S", "Cb", "Cbp", "D", "Dp", "Hd", "Hp", "P", "Pp", "K", "mode")]


# name -> (restype, argtypes); every symbol the header declares
SIGNATURES = {
    "pasn_version": (c_int, []),
    "pasn_comm_unique_id": (c_int, [c_void_p]),
    "pasn_comm_init": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "pasn_allreduce": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "pasn_comm_destroy": (c_int, [c_void_p]),
    "pasn_last_error": (c_char_p, []),
    "pasn_tuning_reload": (None, []),
    "pasn_tuning_get": (c_char_p, [c_char_p]),
    "pasn_tuning_report": (c_int, [c_char_p, c_int, c_int]),
    "pasn_first_conv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_void_p]),
    "pasn_first_conv_gray_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_float, c_float, c_void_p]),
    "pasn_first_conv_mfma_slot": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "pasn_first_conv_mfma_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_float, c_float, c_void_p]),
    "pasn_x3d_stem_mfma_supported": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "pasn_x3d_stem_mfma_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_float, c_float, c_void_p]),
    "pasn_x3d_stem_gray_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_float, c_float, c_void_p]),
    "pasn_x3d_stem_supported": (c_int, [POINTER(ConvDesc)]),
    "pasn_x3d_stem_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_void_p]),
    "pasn_conv3d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_conv3d_pair_supported": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc), c_int]),
    "pasn_conv3d_pair_variant": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc), c_int, c_int]),
    "pasn_conv3d_pair_se_supported": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc), c_int, c_int]),
    "pasn_conv3d_pair_se_fwd": (c_int, [c_void_p] * 6 + [c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p, POINTER(ConvDesc)] + [c_void_p] * 4
                                + [POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_conv3d_pair_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc),
                                     c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_conv3d_variant": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "pasn_conv3d_se_supported": (c_int, [POINTER(ConvDesc), c_int, c_int, c_int]),
    "pasn_conv3d_se_fwd": (c_int, [c_void_p] * 6 + [c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_conv3d_short_supported": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc), c_int]),
    "pasn_conv3d_short_fwd": (c_int, [c_void_p] * 9 + [POINTER(ConvDesc), POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_x3d_edp_supported": (c_int, [POINTER(ConvDesc)] * 4 + [c_int]),
    "pasn_x3d_edp_fwd": (c_int, [c_void_p] * 15 + [POINTER(ConvDesc)] * 4 + [c_int, c_void_p]),
    "pasn_x3d_pe_supported": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc), c_int, c_int]),
    "pasn_x3d_pe_fwd": (c_int, [c_void_p] * 6 + [c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p, POINTER(ConvDesc)] + [c_void_p] * 4
                        + [POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_dwconv3d_pool_blocks": (c_int, [POINTER(ConvDesc), c_int]),
    "pasn_dwconv3d_se_pool_blocks": (c_int, [POINTER(ConvDesc), c_int]),
    "pasn_dwconv3d_variant": (c_int, [POINTER(ConvDesc), c_int]),
    "pasn_dwconv3d_se_supported": (c_int, [POINTER(ConvDesc), c_int, c_int]),
    "pasn_dwconv3d_se_fwd": (c_int, [c_void_p] * 6 + [POINTER(ConvDesc), c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p]),
    "pasn_dwconv3d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_se_gate_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pasn_maxpool3d_fwd": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_l2_head_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "pasn_xproto_head_splits": (c_int, [POINTER(XProtoDesc)]),
    "pasn_xproto_head_workspace_bytes": (c_size_t, [POINTER(XProtoDesc), c_int]),
    "pasn_xproto_head_fwd": (c_int, [c_void_p] * 17 + [POINTER(XProtoDesc), c_int, c_void_p]),
    "pasn_x3d_expdw_supported": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc), c_int]),
    "pasn_x3d_expdw_variant": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc), c_int]),
    "pasn_x3d_expdw_pool_blocks": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc), c_int]),
    "pasn_x3d_expdw_fwd": (c_int, [c_void_p] * 9 + [POINTER(ConvDesc), POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_xproto_chain_supported": (c_int, [POINTER(XProtoDesc), c_int]),
    "pasn_xproto_chain_workspace_bytes": (c_size_t, [POINTER(XProtoDesc)]),
    "pasn_xproto_chain_fwd": (c_int, [c_void_p] * 17 + [POINTER(XProtoDesc), c_int, c_void_p]),
    "pasn_push_xproto_update": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int64, c_void_p]),
    # ---- training path
    "pasn_train_chunks": (c_int, [c_int, c_int, c_int]),
    "pasn_se_gate_bwd_stat": (c_int, [c_void_p] * 17 + [c_int] * 5 + [c_void_p]),
    "pasn_bn_bwd_apply_se": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    "pasn_pack_chunk": (c_int, []),
    "pasn_pack_weights": (c_int, [c_void_p] * 3 + [c_int, c_void_p]),
    "pasn_dwconv3d_stats_rows": (c_int, [POINTER(ConvDesc), c_int]),
    "pasn_dwconv3d_dgrad_reduce_rows": (c_int, [POINTER(ConvDesc), c_int]),
    "pasn_dwconv3d_dgrad_reduce": (c_int, [c_void_p] * 7 + [c_int] + [c_void_p] * 4 + [POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_dwconv3d_stats_fwd": (c_int, [c_void_p] * 10 + [c_float, c_float, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_bn_stats_fwd": (c_int, [c_void_p] * 6 + [c_float, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pasn_affine_act_fwd": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "pasn_unit_bwd_reduce": (c_int, [c_int] + [c_void_p] * 10 + [c_int] * 6 + [c_void_p]),
    "pasn_bn_bwd_apply": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    # ... with statistics groups (stat [groups][4][Cp], coef [groups][2][Cp]): the paired training pass
    "pasn_bn_stats_fwd_g": (c_int, [c_void_p] * 6 + [c_float, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pasn_dwconv3d_stats_fwd_g": (c_int, [c_void_p] * 10 + [c_float, c_float, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_void_p]),
    "pasn_affine_act_fwd_g": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "pasn_unit_bwd_reduce_g": (c_int, [c_int] + [c_void_p] * 10 + [c_int] * 7 + [c_void_p]),
    "pasn_bn_bwd_apply_g": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "pasn_se_gate_bwd_stat_g": (c_int, [c_void_p] * 17 + [c_int] * 6 + [c_void_p]),
    "pasn_bn_bwd_apply_se_g": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    "pasn_se_bwd_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "pasn_se_gate_bwd": (c_int, [c_void_p] * 12 + [c_int] * 5 + [c_void_p]),
    "pasn_scatter_strided": (c_int, [c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_void_p]),
    "pasn_add_inplace": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "pasn_maxpool3d_bwd": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_conv3d_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_conv3d_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc), c_int]),
    "pasn_conv3d_wgrad_ws": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p, c_void_p]),
    "pasn_first_conv_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc), c_int]),
    "pasn_first_conv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_int, c_void_p, c_void_p]),
    "pasn_dwconv3d_wgrad_workspace_floats": (c_size_t, [POINTER(ConvDesc)]),
    "pasn_dwconv3d_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_dwconv3d_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(ConvDesc), c_int, c_void_p]),
    "pasn_xproto_tail_fwd": (c_int, [c_void_p] * 8 + [POINTER(XProtoDesc), c_int, c_void_p]),
    "pasn_xproto_tail_workspace_bytes": (c_size_t, [POINTER(XProtoDesc)]),
    "pasn_xproto_tail_fwd_ws": (c_int, [c_void_p] * 8 + [POINTER(XProtoDesc), c_int, c_void_p, c_void_p]),
    "pasn_xproto_tail_bwd": (c_int, [c_void_p] * 14 + [POINTER(XProtoDesc), c_int, c_void_p]),
    "pasn_l2_head_bwd": (c_int, [c_void_p] * 11 + [c_int] * 8 + [c_float, c_void_p]),
    "pasn_affine_warp_fwd": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_float, c_int, c_void_p]),
    "pasn_affine_warp_bwd": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_float, c_void_p]),
    "pasn_push_ppnet_update": (c_int, [c_void_p] * 4 + [c_int] + [c_void_p] * 3 + [c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """The loaded library; raises if it has not been built (``python __graft_entry__.py`` builds it)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().pasn_last_error().decode("utf-8", "replace")
        if rc == 1:
            raise ValueError(f"protoasnet_amd: {msg}")
        raise RuntimeError(f"protoasnet_amd (status {rc}): {msg}")


def dtype_code(dtype) -> int:
    import torch

    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.uint8:
        return U8  # input clips of the grey first-layer entry points only
    raise TypeError(f"protoasnet_amd kernels compute in float32 or bfloat16, not {dtype}")


def ptr(t) -> int:
    """Raw device pointer of a tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream


# ---- tuning switches (csrc/tuning.h): the library reads the PASN_* environment ONCE ----------------------------------------------------
_TUNING_EPOCH = [0]


def tuning_reload() -> None:
    """Take a new snapshot of the PASN_* environment (after ``os.environ`` / ``monkeypatch.setenv`` changed it: tests, A/B tools).
    Bumps ``tuning_epoch()``: compiled plans, training runners and captured graphs are keyed on it, because their workspaces were sized
    under the geometry switches of the snapshot they were built with (PASN_TRAIN_BLOCKS, PASN_DWWG_BLOCKS ...) while the launchers
    recompute the geometry per launch -- a plan from an older snapshot is rebuilt instead of replayed."""
    lib().pasn_tuning_reload()
    _TUNING_EPOCH[0] += 1


def tuning_epoch() -> int:
    return _TUNING_EPOCH[0]


def tuning_get(name: str):
    """A registered switch's value in the library's snapshot, or None.  The Python host side routes on the same snapshot."""
    v = lib().pasn_tuning_get(name.encode())
    return None if v is None else v.decode()


def tuning_report(with_registry: bool = False) -> str:
    n = lib().pasn_tuning_report(None, 0, int(with_registry))
    buf = ctypes.create_string_buffer(n + 1)
    lib().pasn_tuning_report(buf, n + 1, int(with_registry))
    return buf.value.decode()


class tuning_env:
    """``with tuning_env(PASN_X="0", PASN_Y=None): ...`` -- set / unset switches for a block, snapshot reloaded on entry and exit."""

    def __init__(self, **kv):
        self.kv, self.saved = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.saved[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        tuning_reload()
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        tuning_reload()
        return False
