"""ctypes binding of libtt_hotpath.so (C ABI in include/tt_hotpath.h).

The library is the product path: if it cannot be loaded this module raises --
there is no CPU or eager-PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# TT_HOTPATH_LIB: load another build of the SAME library (tools/mips_variants.sh A/Bs compile-time kernel variants)
LIB_PATH = os.environ.get("TT_HOTPATH_LIB") or os.path.join(_PKG, "lib", "libtt_hotpath.so")

TT_GEMM_NT, TT_GEMM_NN, TT_GEMM_TN = 0, 1, 2
TT_EPI_NONE, TT_EPI_RELU, TT_EPI_RELU_MASK = 0, 1, 2
TT_F32, TT_BF16, TT_F16X2 = 0, 1, 2
TT_E_BADARG, TT_E_WORKSPACE, TT_E_UNSUPPORTED = -1, -2, -3
TT_DEBIAS_COMBINED, TT_DEBIAS_POSITION, TT_DEBIAS_USER = 0, 1, 2
TT_COMM_ID_BYTES = 128
TT_COMM_F32, TT_COMM_I32, TT_COMM_I64, TT_COMM_U8 = 0, 1, 2, 3
TT_COMM_SUM, TT_COMM_MAX = 0, 1
TT_MAX_GRAD_SOURCES = 4
ABI_VERSION = 4

_vp, _i64, _i32, _int = C.c_void_p, C.c_int64, C.c_int32, C.c_int


class GradSources(C.Structure):
    """tt_grad_sources (host struct, passed by pointer)."""

    _fields_ = [
        ("rows", _vp * TT_MAX_GRAD_SOURCES),
        ("ld", _i64 * TT_MAX_GRAD_SOURCES),
        ("first", _i64 * (TT_MAX_GRAD_SOURCES + 1)),
        ("n_sources", _i32),
    ]


class AdamTensor(C.Structure):
    """tt_adam_tensor."""

    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("n", _i64)]


class AdamStashJob(C.Structure):
    """tt_adam_stash_job."""

    _fields_ = [("W", _vp), ("M", _vp), ("V", _vp), ("n_rows", _i64), ("dim", _i64), ("ids", _vp), ("n_ids", _i64),
                ("side", _vp), ("side_bytes", _i64)]


class AdamFinishJob(C.Structure):
    """tt_adam_finish_job."""

    _fields_ = [("W", _vp), ("M", _vp), ("V", _vp), ("n_rows", _i64), ("dim", _i64), ("src", C.POINTER(GradSources)),
                ("n_ids", _i64), ("sorted_ids", _vp), ("perm", _vp), ("seg_begin", _vp), ("n_unique", _vp), ("side", _vp),
                ("side_bytes", _i64)]


class TowerFwdSide(C.Structure):
    """tt_tower_fwd_side."""

    _fields_ = [("table", _vp), ("n_rows", _i64), ("ids", _vp), ("feats", _vp), ("ldf", _i64), ("F", _i64), ("W1", _vp), ("b1", _vp),
                ("W2", _vp), ("b2", _vp), ("W3", _vp), ("b3", _vp), ("y", _vp), ("ldy", _i64), ("h_out", _vp), ("tin_out", _vp)]


class TowerBwdSide(C.Structure):
    """tt_tower_bwd_side."""

    _fields_ = [("dy", _vp), ("ldy", _i64), ("W2", _vp), ("W3", _vp), ("h", _vp), ("d_emb", _vp), ("ld_demb", _i64), ("d_f", _vp),
                ("dh", _vp)]


class TowerWgradSide(C.Structure):
    """tt_tower_wgrad_side."""

    _fields_ = [("dy", _vp), ("ldy", _i64), ("tin", _vp), ("d_f", _vp), ("h", _vp), ("dh", _vp), ("feats", _vp), ("ldf", _i64),
                ("F", _i64), ("dW1", _vp), ("db1", _vp), ("dW2", _vp), ("db2", _vp), ("dW3", _vp), ("db3", _vp), ("ws", _vp),
                ("ws_bytes", _i64)]


class PlanJob(C.Structure):
    """tt_plan_job."""

    _fields_ = [("ids", _vp), ("n_ids", _i64), ("n_rows", _i64), ("sorted_ids", _vp), ("perm", _vp), ("seg_begin", _vp), ("n_unique", _vp)]


TT_PLAN_MAX_JOBS = 4
TT_ROUTE_MAX_JOBS = 8


class RouteJob(C.Structure):
    """tt_route_job."""

    _fields_ = [("ids", _vp), ("n_ids", _i64), ("n_rows", _i64), ("rows_per_rank", _i64), ("counts", _vp), ("max_count", _vp),
                ("ws", _vp), ("ws_bytes", _i64), ("cap", _i64), ("send_ids", _vp), ("slot_of", _vp), ("src_of", _vp)]


class RouteServeJob(C.Structure):
    """tt_route_serve_job."""

    _fields_ = [("ids", _vp), ("n_ids", _i64), ("lo", _i64), ("n_local", _i64), ("local", _vp), ("table", _vp), ("dtype", _int),
                ("dim", _i64), ("rows", _vp)]


# name -> (restype, argtypes); mirrors include/tt_hotpath.h declaration by declaration
SIGNATURES = {
    "tt_abi_version": (_int, []),
    "tt_last_error_string": (C.c_char_p, []),
    "tt_profile_enable": (_int, [_int]),
    "tt_profile_read": (_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "tt_profile_filter": (_int, [C.c_char_p]),
    "tt_profile_pause": (_int, [_int]),
    "tt_gather_rows": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "tt_gemm_workspace_bytes": (_i64, [_int, _i64, _i64, _i64]),
    "tt_gemm_f32": (_int, [_int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _int, _vp,
                           _i64, _int, _vp, _i64, _vp]),
    "tt_gemm_tn_colsum_f32": (_int, [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _vp, _i64, _vp]),
    "tt_colsum_workspace_bytes": (_i64, [_i64, _i64]),
    "tt_colsum_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp]),
    "tt_tower_supported": (_int, [_i64, _i64, _i64, _i64]),
    "tt_tower_fwd": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64,
                            _vp, _vp, _vp, _vp]),
    "tt_tower_bwd_data": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "tt_tower_bwd_weights_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "tt_tower_bwd_weights": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _i64, _vp]),
    "tt_adam_begin_ids": (_int, [_vp, _vp, _i64, C.POINTER(AdamStashJob), _i32, _vp]),
    "tt_adam_begin_ids_planes": (_int, [_vp, _vp, _i64, C.POINTER(AdamStashJob), _i32, _i32, _vp]),
    "tt_adam_tables_finish": (_int, [C.POINTER(AdamFinishJob), _i32, _vp, _vp]),
    "tt_inbatch_ce_fwd_du_loss": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                                         _vp, _vp, _i64, _vp]),
    "tt_scale_rows_g": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "tt_tower_x_supported": (_int, [_i64, _i64, _i64, _i64, _i64]),
    "tt_tower_fwd_x": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                              _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "tt_tower_bwd_data_x": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp]),
    "tt_tower_bwd_weights_x_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64, _i64]),
    "tt_tower_bwd_weights_x": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tt_inbatch_ce_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "tt_inbatch_ce_fwd": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "tt_inbatch_ce_bwd": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp,
                                 _i64, _vp, _i64, _vp]),
    "tt_inbatch_ce_fwd_du": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "tt_inbatch_ce_logits_bytes": (_i64, [_i64, _i64]),
    "tt_inbatch_ce_fwd_du_keep": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64,
                                         _vp, _i64, _vp]),
    "tt_inbatch_ce_bwd_kept": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "tt_ce16_supported": (_int, [_i64, _i64, _i64]),
    "tt_ce16_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "tt_ce16_fwd_du_keep": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "tt_ce16_bwd_kept": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "tt_ce16_bwd_recompute": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _int, _vp]),
    "tt_scale_rows": (_int, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp]),
    "tt_weighted_mean_loss": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tt_value_weights": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "tt_weighted_loss_global": (_int, [_vp, _vp, _vp, _i64, C.c_float, _vp, _vp, _vp]),
    "tt_debias_loss_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "tt_debias_loss_fwd": (_int, [_int, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64,
                                  _vp, _vp]),
    "tt_debias_loss_bwd": (_int, [_int, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp,
                                  _vp, _vp]),
    "tt_rowgrad_workspace_bytes": (_i64, [_i64]),
    "tt_rowgrad_plan": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tt_rowgrad_dense": (_int, [C.POINTER(GradSources), _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tt_adam_advance": (_int, [_vp, _vp]),
    "tt_adam_table_workspace_bytes": (_i64, [_i64, _i64]),
    "tt_adam_table": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, C.POINTER(GradSources), _i64, _vp, _vp, _vp,
                             _vp, _vp, _i64, _vp]),
    "tt_adam_table_stash": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tt_adam_table_stash_ids": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "tt_adam_table_sweep": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "tt_adam_tables_sweep": (_int, [C.POINTER(AdamTensor), _i32, _vp, _i32, _vp]),
    "tt_adam_marks_words": (_i64, [_i64]),
    "tt_adam_marked_supported": (_int, [_i64]),
    "tt_adam_mark_rows": (_int, [_vp, _i64, _i64, _vp, _i64, _vp]),
    "tt_adam_tables_sweep_marked": (_int, [C.POINTER(AdamTensor), C.POINTER(_i64), C.POINTER(_vp), _i32, _vp, _i32, _vp]),
    "tt_adam_table_finish": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, C.POINTER(GradSources), _i64, _vp, _vp, _vp,
                                    _vp, _vp, _i64, _vp]),
    "tt_adam_advance_tab": (_int, [_vp, _vp, _i64, _vp]),
    "tt_adam_rows_catchup": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "tt_adam_table_lazy": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, C.POINTER(GradSources), _i64, _vp, _vp, _vp, _vp,
                                  _vp, _i64, _vp, _vp, _i64, _vp]),
    "tt_adam_table_flush": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "tt_stream_create_low_priority": (_int, [C.POINTER(_vp)]),
    "tt_stream_destroy": (_int, [_vp]),
    "tt_adam_dense": (_int, [C.POINTER(AdamTensor), _i32, _vp, _vp]),
    "tt_pack_grads": (_int, [C.POINTER(AdamTensor), _i32, _vp]),
    "tt_copy_buffers": (_int, [C.POINTER(AdamTensor), _i32, _vp]),
    "tt_hist_embed_pool": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp]),
    "tt_hist_dx_pool_bwd": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp]),
    "tt_hist_pool_bwd": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "tt_attn_fwd": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "tt_attn_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "tt_enc_last_supported": (_int, [_i64, _i64, _i64]),
    "tt_enc_last_fwd": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tt_enc_last_bwd_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64]),
    "tt_enc_last_bwd": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _i64, _vp]),
    "tt_enc_last_bwd_data": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tt_enc_last_bwd_weights": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tt_stream_copy": (_int, [_vp, _vp, _i64, _vp]),
    "tt_mfma_probe_flops": (_i64, [_int, _i32]),
    "tt_mfma_probe": (_int, [_int, _i32, _vp, _i64, _vp]),
    "tt_tower_fwd_pair": (_int, [C.POINTER(TowerFwdSide), _i64, _i64, _i64, _vp, _vp]),
    "tt_tower_bwd_data_pair": (_int, [C.POINTER(TowerBwdSide), _i64, _i64, _i64, _vp]),
    "tt_tower_bwd_weights_pair": (_int, [C.POINTER(TowerWgradSide), _i64, _i64, _i64, _vp]),
    "tt_rowgrad_plan_jobs_supported": (_int, [_i64]),
    "tt_rowgrad_plan_jobs": (_int, [C.POINTER(PlanJob), _i32, _vp, _vp]),
    "tt_route_workspace_bytes": (_i64, [_i64, _i32]),
    "tt_route_count": (_int, [_vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "tt_route_build": (_int, [_vp, _i64, _i64, _i64, _i32, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "tt_route_localize": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "tt_route_count_jobs": (_int, [C.POINTER(RouteJob), _i32, _i32, _vp, _vp]),
    "tt_route_build_jobs": (_int, [C.POINTER(RouteJob), _i32, _i32, _vp, _vp]),
    "tt_route_serve_jobs": (_int, [C.POINTER(RouteServeJob), _i32, _vp]),
    "tt_comm_unique_id": (_int, [_vp]),
    "tt_comm_init": (_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "tt_comm_destroy": (_int, [_vp]),
    "tt_comm_size": (_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "tt_comm_alltoall": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "tt_comm_allgather": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "tt_comm_reduce_scatter": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp]),
    "tt_comm_allreduce": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp]),
    "tt_comm_broadcast": (_int, [_vp, _vp, _i64, _int, _i32, _vp]),
    "tt_mips_workspace_bytes": (_i64, [_i64, _i64, _i64, _i64, _int]),
    "tt_mips_topk": (_int, [_vp, _vp, _int, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "tt_mips_merge_workspace_bytes": (_i64, [_i64, _i64]),
    "tt_mips_merge": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "tt_f32_to_bf16": (_int, [_vp, _vp, _i64, _vp]),
    "tt_mips_split_rows": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "tt_mips_unscale": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "tt_gather_rows_bf16": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


class NativeLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library once and bind every declared entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m two_tower_models_amd.build` "
            "(hipcc, gfx950).  two_tower_models_amd has no fallback path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover - build/ABI mismatch
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.tt_abi_version()
    if got != ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: library {got}, bindings {ABI_VERSION}")
    _lib = lib
    return lib


# tests/ only: when a list, receives -- in host order -- the name of every library call that enqueued work (`check`) and
# "issue:<tag>" / "wait:<tag>" for every exchange of collectives.py: the ORDER in which a step issues its kernels and its
# collectives is what decides which kernels can cover which exchange, and it can be asserted without a second GPU.
trace: Optional[list] = None


def check(rc: int, what: str) -> None:
    if trace is not None:
        trace.append(what)
    if rc != 0:
        msg = load().tt_last_error_string().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """hipStream_t of torch's current stream (kernels run stream-ordered with torch ops).  Called once per kernel
    launch: the raw-handle query is ~10x cheaper than building a torch.cuda.Stream object each time (25 calls per step
    were 0.2 ms of host time at C2, where the host enqueue rate bounds the step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "two_tower_models_amd runs on MI355X only: got a CPU tensor "
                "(move the module and its inputs to the GPU; there is no CPU path)"
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


_streams = {}


def low_priority_stream(device: torch.device) -> torch.cuda.Stream:
    """Least-priority HIP stream (created by the library), wrapped for torch's event API.  ONE per device and process:
    ROCm multiplexes HIP streams onto a handful of hardware queues in creation order, so every further stream shifts
    which streams share a queue -- the second optimiser of a process once got a sort stream that shared its queue with
    the sweep (C2: 1.75 instead of 1.37 ms per step).  Optimisers are stream-ordered on their main stream anyway."""
    key = ("low", device.index)
    if key not in _streams:
        with torch.cuda.device(device):
            raw = _vp()
            check(load().tt_stream_create_low_priority(C.byref(raw)), "tt_stream_create_low_priority")
        _streams[key] = torch.cuda.ExternalStream(raw.value, device=device)
    return _streams[key]


def aux_stream(device: torch.device) -> torch.cuda.Stream:
    """The third stream (row-plan sorts next to the backward): one per device and process, see low_priority_stream."""
    key = ("aux", device.index)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=device)
    return _streams[key]


# ----------------------------------------------------------------- scratch
class _Scratch:
    """One growable byte buffer per (device, slot).  Calls are stream-ordered on the
    current stream, so a buffer can be reused by the next call as soon as the previous
    launch has been enqueued."""

    def __init__(self):
        self._bufs = {}
        self._captured = set()  # (device, slot) buffers whose address a captured hipGraph has baked in
        self._retired = []  # such buffers after they were outgrown: kept alive for the graph's replays

    def get(self, dev: torch.device, nbytes: int, slot: str = "ws") -> torch.Tensor:
        key = (dev.index, slot)
        buf = self._bufs.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if buf is None or buf.numel() < nbytes:
            if capturing:
                raise RuntimeError("scratch buffer would grow during graph capture; run a warm-up step first")
            if buf is not None and key in self._captured:
                # an eager call after a capture needs more room: the graph keeps replaying into the old
                # buffer, so it must never go back to the allocator
                self._retired.append(buf)
                self._captured.discard(key)
            buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=dev)
            self._bufs[key] = buf
        if capturing:
            self._captured.add(key)
        return buf


scratch = _Scratch()


class _OobFlags:
    """Device-side out-of-range-id flag (one int32 per device) with a non-blocking
    host mirror, so the training loop never synchronises for it.  `raise_if_set`
    reports a bad id at the latest one call after the kernel that saw it."""

    def __init__(self):
        self._dev = {}
        self._host = {}
        self._event = {}

    def flag(self, dev: torch.device) -> torch.Tensor:
        f = self._dev.get(dev.index)
        if f is None:
            f = torch.zeros(1, dtype=torch.int32, device=dev)
            self._dev[dev.index] = f
            self._host[dev.index] = torch.zeros(1, dtype=torch.int32).pin_memory()
        return f

    def poll(self, dev: torch.device, blocking: bool = False) -> None:
        if torch.cuda.is_current_stream_capturing():
            return
        f = self.flag(dev)
        host = self._host[dev.index]
        ev = self._event.get(dev.index)
        if blocking:
            if int(f.item()) != 0:
                f.zero_()
                raise IndexError("index out of range in self")  # torch's nn.Embedding message
            return
        if ev is not None and ev.query():
            if int(host[0]) != 0:
                f.zero_()
                host.zero_()
                self._event.pop(dev.index, None)
                raise IndexError("index out of range in self")
        host.copy_(f, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._event[dev.index] = ev


oob = _OobFlags()
