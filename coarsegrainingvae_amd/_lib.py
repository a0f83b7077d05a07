"""ctypes binding of libcgvae_hip.so (C ABI: include/cgvae_hip.h).

There is deliberately no fallback: if the shared library is missing or a call fails, the
product path raises.  Only raw device pointers, sizes and the current HIP stream cross the
boundary; torch is used for allocation and stream bookkeeping only.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import threading

import torch

from . import ktimer

_PKG = os.path.dirname(os.path.abspath(__file__))
# CGV_LIB: path of ANOTHER BUILD of the same library (an instrumented or A/B variant made by tools/build_variant.sh) to load
# in place of the shipped one -- so that no tool ever has to copy a variant over the shipped file.  Still no fallback:
# whatever this names must exist and export every symbol of include/cgvae_hip.h.
LIB_PATH = os.environ.get("CGV_LIB") or os.path.join(_PKG, "libcgvae_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "cgvae_hip.h")

_lock = threading.Lock()
_lib = None

_p, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol the header declares (tests check it)
PROTOTYPES = {
    "cgv_version": (_i, []),
    "cgv_last_error_string": (C.c_char_p, []),
    "cgv_timestamp": (_i, [_p, _p]),
    "cgv_timestamp_hz": (_i, []),
    "cgv_sustained_clock_probe": (_i, [_p, _p, _i, _i, _p]),
    "cgv_set_option": (_i, [_i, _i]),
    "cgv_get_option": (_i, [_i]),
    "cgv_reset_options": (_i, []),
    "cgv_rbf_supported": (_i, [_i]),
    "cgv_geom_stride": (_i, [_i]),
    "cgv_geom_unit_offset": (_i, [_i]),
    "cgv_radius_graph_count": (_i, [_p, _p, _i, _i, _f, _i, _p, _p, _p]),
    "cgv_radius_graph_emit": (_i, [_p, _p, _i, _i, _f, _i, _p, _p, _p]),
    "cgv_csr_workspace_bytes": (_sz, [_i]),
    "cgv_csr_build": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "cgv_group_plan_workspace_bytes": (_sz, [_i]),
    "cgv_group_plan_build": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "cgv_group_plan_radix_workspace_bytes": (_sz, [_i]),
    "cgv_group_plan_build_radix": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "cgv_geom_group_stride": (_i, [_i]),
    "cgv_geom_group_unit_offset": (_i, [_i]),
    "cgv_edge_geometry_grouped": (_i, [_p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p]),
    "cgv_edge_geometry": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p]),
    "cgv_plan_jobs_max": (_i, []),
    "cgv_plan_job_bytes": (_i, []),
    "cgv_plan_jobs_build": (_i, [_p, _i, _p]),
    "cgv_geom_jobs_max": (_i, []),
    "cgv_geom_job_bytes": (_i, []),
    "cgv_geom_jobs_build": (_i, [_p, _i, _p]),
    "cgv_segment_reduce": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "cgv_embedding_rows": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "cgv_embedding_rows2": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _p, _i, _p]),
    "cgv_segment_reduce_pair": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p]),
    "cgv_segment_broadcast": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "cgv_equi_msg_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, C.c_int64, C.c_int64, _p, _p, _p]),
    "cgv_equi_msg_grouped_supported": (_i, [_i, _i, _i]),
    "cgv_equi_msg_fwd_grouped": (_i, [_p] * 9 + [_i, _i, _i, _i, C.c_int64, C.c_int64, _p, _p, _p]),
    "cgv_equi_msg_grouped_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "cgv_equi_msg_fwd_grouped_parts": (_i, [_p] * 9 + [_i, _i, _i, _i, C.c_int64, C.c_int64, _p, _p, _i, _p, C.c_size_t, _p]),
    "cgv_equi_msg_balanced_supported": (_i, [_i, _i, _i]),
    "cgv_equi_msg_balanced_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "cgv_equi_msg_fwd_balanced": (_i, [_p] * 10 + [_i, _i, _i, _i, C.c_int64, _p, _p, _p, C.c_size_t, _p]),
    "cgv_equi_msg_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "cgv_equi_msg_bwd": (_i, [_p] * 13 + [_i, _i, _i, C.c_int64, C.c_int64, _p, _sz, _p]),
    "cgv_pseudo_msg_fwd": (_i, [_p] * 14 + [_i, _i, _i, _i, _p]),
    "cgv_pseudo_msg_fwd_rows": (_i, [_p] * 15 + [_i, _i, _i, _i, C.c_int64, _p]),
    "cgv_pseudo_msg_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "cgv_pseudo_msg_bwd": (_i, [_p] * 24 + [_i, _i, _i, _i, C.c_int64, _p, _sz, _p]),
    "cgv_pseudo_msg_bwd_deferred": (_i, [_p] * 22 + [_i, _i, _i, _i, C.c_int64, _p, _sz, _p, _p]),
    "cgv_update_rows_from_vec": (_i, [_p, _p, _i, _i, _p]),
    "cgv_tile_bwd_input_plan": (_i, [_i, _i, _i, _i, _p, _p]),
    "cgv_tile_linear_bwd_input_norm_stack": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p]),
    "cgv_update_vec_from_rows": (_i, [_p, _p, _p, _i, _i, _p]),
    "cgv_update_norm_stack_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "cgv_update_norm_stack_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "cgv_update_gate_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "cgv_update_gate_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "cgv_skinny_max_rows": (_i, []),
    "cgv_skinny_supported": (_i, [_i, _i, _i]),
    "cgv_skinny_fwd_supported": (_i, [_i, _i, _i]),
    "cgv_skinny_linear_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "cgv_skinny_bwd_input_supported": (_i, [_i, _i, _i]),
    "cgv_skinny_bwd_input_workspace_bytes": (_sz, [_i, _i, _i]),
    "cgv_skinny_linear_bwd_input": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "cgv_skinny_bwd_input_plan": (_i, [_i, _i, _i, _p, _p]),
    "cgv_skinny_linear_bwd_input_slices": (_i, [_p, _p, _i, C.c_int64, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _p]),
    "cgv_slice_sum": (_i, [_p, _p, _i, C.c_int64, _p, C.c_int64, _p]),
    "cgv_update_vec_from_rows_slices": (_i, [_p, _i, C.c_int64, _p, _p, _i, _i, _p]),
    "cgv_update_norm_stack_bwd_slices": (_i, [_p, _i, C.c_int64, _p, _p, _p, _p, _i, C.c_int64, _p, _p, _i, _i, _i, _i, _p]),
    "cgv_update_gate_bwd_slices": (_i, [_p, _p, _p, _p, _p, _i, C.c_int64, _p, _p, _p, _p, _i, _i, _i, _p]),
    "cgv_decoder_layer_supported": (_i, [_i, _i, _i]),
    "cgv_decoder_slice_floats": (C.c_int64, [_i, _i]),
    "cgv_batch_load_rows": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _p]),
    "cgv_reparam_sample": (_i, [_p, _p, _p, _p, C.c_int64, _p, _p]),
    "cgv_reparam_bwd": (_i, [_p, _p, _p, _p, _p, _p, C.c_int64, _p]),
    "cgv_pair_linear_fwd": (_i, [_p] * 10 + [_i] * 5 + [_p]),
    "cgv_equi_msg_bwd_deferred": (_i, [_p] * 11 + [_i, _i, _i, C.c_int64, C.c_int64, _p, C.c_size_t, _p, _p, _p]),
    "cgv_filter_reduce_jobs_max": (_i, []),
    "cgv_filter_reduce_job_bytes": (_i, []),
    "cgv_filter_reduce_jobs": (_i, [_p, _i, _p]),
    "cgv_pair_linear_bwd_input": (_i, [_p] * 6 + [_i, _i, _p, _p, _i, _i, _i, _p, C.c_size_t, _p]),
    "cgv_loss_tail_supported": (_i, [_i, _i, _i, _i]),
    "cgv_loss_tail_workspace_bytes": (_sz, [_i]),
    "cgv_loss_tail": (_i, [_p] * 12 + [_i, _i, _i, _i, _i, _f, _f] + [_p] * 9 + [_p, _sz, _p]),
    "cgv_multi_linear_max": (_i, []),
    "cgv_multi_linear_fwd": (_i, [_i] + [_p] * 6 + [_i, _i, _i, _p]),
    "cgv_multi_linear_bwd_input": (_i, [_i, _i] + [_p] * 5 + [_i, _i, _i, _p, C.c_size_t, _p]),
    "cgv_decoder_max_edges": (_i, []),
    "cgv_decoder_block_channels": (_i, [_i]),
    "cgv_decoder_debug_clock": (_i, [_p]),
    "cgv_decoder_msg_fwd": (_i, [_p] * 18 + [_i, _i, _i, _i, _p]),
    "cgv_decoder_dense_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "cgv_prior_msg_fwd": (_i, [_p] * 13 + [_i, _i, _i, _i, _i, _p]),
    "cgv_prior_msg_bwd": (_i, [_p] * 8 + [_i, C.c_int64] + [_p] * 6 + [C.c_int64, _i, _i, _i, _i, _p]),
    "cgv_decoder_uv_fwd": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "cgv_decoder_gate_fwd": (_i, [_p] * 9 + [_i, _i, _p]),
    "cgv_update_rows_fused_supported": (_i, [_i, _i]),
    "cgv_update_uv_norm_fwd_fused": (_i, [_p] * 5 + [_i, _i, _p]),
    "cgv_update_gate_fwd_fused": (_i, [_p] * 9 + [_i, _i, _p]),
    "cgv_decoder_gate_bwd": (_i, [_p, _p, _p, _p, _i, C.c_int64, _p, _p, _p, _p, _p, _p, C.c_int64, _i, _i, _p]),
    "cgv_decoder_dense_bwd": (_i, [_p, _i, C.c_int64, _p, _i, _p, _p, _p, C.c_int64, _i, _i, _i, _p]),
    "cgv_decoder_uv_bwd": (_i, [_p, _i, C.c_int64, _p, _p, _p, _p, _p, _p, _p, _p, C.c_int64, _i, _i, _p]),
    "cgv_decoder_msg_bwd": (_i, [_p] * 15 + [_p, _i, C.c_int64, _p, _p, _p] + [_p] * 8 + [C.c_int64, _i, _i, _i, _i, _p]),
    "cgv_decoder_slices_to_dense": (_i, [_p, _p, _i, C.c_int64, _p, _i, _i, _p]),
    "cgv_dense_grad_prepare": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "cgv_tile_supported": (_i, [_i, _i, _i]),
    "cgv_tile_linear_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "cgv_tile_linear_bwd_input": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "cgv_tile_linear_bwd_input_act": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "cgv_tile_linear_bwd_input_act_add": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "cgv_tile_linear_bwd_input_act_add_bcast": (_i, [_p] * 7 + [_i, _p, _i, _i, _i, _i, _p]),
    "cgv_tile_pair_supported": (_i, [_i, _i, _i]),
    "cgv_tile_pair_linear_fwd": (_i, [_p] * 10 + [_i, _i, _i, _i, _i, _p]),
    "cgv_tile_pair_linear_bwd_input": (_i, [_p] * 10 + [_i, _i, _i, _i, _i, _p]),
    "cgv_tile_linear_bwd_input_out": (_i, [_p] * 5 + [_i, _i, _i, _i, _p, _i, _p]),
    "cgv_tile_bwd_input_split": (_i, [_p, _sz, _p]),
    "cgv_tile_pair_linear_bwd_input_out": (_i, [_p] * 10 + [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    "cgv_tile_linear_bwd_input_sum2": (_i, [_p] * 10 + [_i, _p, _i, _i, _i, _i, _i, _p]),
    "cgv_segment_reduce2": (_i, [_p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _p]),
    "cgv_skinny_linear_bwd_input_add": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "cgv_skinny_linear_bwd_input_out": (_i, [_p] * 5 + [_i, _i, _i, _i, _p, _i, _p, _sz, _p]),
    "cgv_tile_linear_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "cgv_wgrad_record_bytes": (_i, []),
    "cgv_wgrad_plan": (_i, [_i, _i, _i, _p, _p, _p]),
    "cgv_wgrad_lds_floats": (_i, [_i, _i]),
    "cgv_grouped_wgrad": (_i, [_p, _i, _i, _i, _p]),
    "cgv_wgrad_gathered_plan": (_i, [_i, _i, _i, _i, _p, _p]),
    "cgv_grouped_wgrad_gathered": (_i, [_p, _i, _i, _p]),
    "cgv_wgrad_gathered_plan_tile": (_i, [_i, _i, _i, _i, _i, _p, _p]),
    "cgv_grouped_wgrad_gathered_tile": (_i, [_p, _i, _i, _i, _p]),
    "cgv_grouped_wgrad_split": (_i, [_p, _i, _i, _p]),
    "cgv_grouped_wgrad_gathered_sumsq": (_i, [_p, _i, _i, _p, _p, _p]),
    "cgv_grouped_wgrad_gathered_adam": (_i, [_p, _i, _i, _p, _p, _p, _p, _f, _f, _f, _f, _p, _p]),
    "cgv_wgrad_strip_max_rows": (_i, []),
    "cgv_wgrad_strip_plan": (_i, [_i, _i, _i, _i, _p]),
    "cgv_wgrad_strip_split_max_rows": (_i, []),
    "cgv_wgrad_strip_split_plane_bytes": (_sz, [_i, _i]),
    "cgv_grouped_wgrad_strip_split": (_i, [_p, _i, _i, _i, _i, _p, _sz, _p]),
    "cgv_grouped_wgrad_strip": (_i, [_p, _i, _i, _i, _p]),
    "cgv_grouped_wgrad_strip_sumsq": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "cgv_grouped_wgrad_strip_adam": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _f, _f, _f, _f, _p, _p]),
    "cgv_pack_record_bytes": (_i, []),
    "cgv_pack_plan": (_i, [_i, _i, _i, _p]),
    "cgv_pack_operands": (_i, [_p, _i, _i, _p]),
    "cgv_reconstruct_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "cgv_reconstruct_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "cgv_elbo_workspace_bytes": (_sz, [_i, _i]),
    "cgv_elbo_fwd": (_i, [_p] * 7 + [_i, _i, _i, _i, _f, _f] + [_p] * 7 + [_p, _sz, _p]),
    "cgv_elbo_scale": (_i, [_p, _p, _p, _p, _p, _i, _p, _i, _p]),
    "cgv_optim_state_floats": (_i, []),
    "cgv_optim_partial_floats": (_i, []),
    "cgv_adam_clip_step": (_i, [_p, _p, _p, _p, C.c_int64, _f, _f, _f, _f, _f, _f, _p, _f, _p, _p, _p]),
    "cgv_optim_prepare": (_i, [_p, C.c_int64, _f, _f, _f, _f, _p, _f, _p, _p, _p]),
    "cgv_adam_apply": (_i, [_p, _p, _p, _p, C.c_int64, _f, _f, _f, _f, _p, _p]),
    "cgv_sgd_apply": (_i, [_p, _p, C.c_int64, _f, _p, _p]),
    "cgv_wgrad_gram": (_i, [_p, _i, _i, _p, _p, C.c_size_t, _p]),
    "cgv_wgrad_gram_workspace_bytes": (C.c_size_t, [_i]),
    "cgv_wgrad_gram_mfma": (_i, [_p, _i, _i, _p, _p, C.c_size_t, _p]),
    "cgv_wgrad_gram_mfma_workspace_bytes": (C.c_size_t, [_i, _i]),
    "cgv_wgrad_gram_mfma_max_rows": (_i, []),
    "cgv_rank_update_supported": (_i, [_i, _i, _i]),
    "cgv_optim_prepare_extra": (_i, [_p, C.c_int64, _p, _i, _f, _f, _f, _f, _p, _f, _p, _p, _p]),
    "cgv_grouped_wgrad_adam": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _f, _f, _f, _f, _p, _p]),
    "cgv_rank_flat_quantum": (_i, []),
    "cgv_rank_flat_plan": (_i, [_i, _i, _i, _i, _p, _p]),
    "cgv_grouped_wgrad_adam_flat": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _f, _f, _f, _p, _p]),
    "cgv_grouped_wgrad_adam_mixed": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _f, _f, _f, _f, _p, _p]),
}


# include/cgvae_hip.h: CGV_OPT_* (A/B switches of the launchers; defaults in csrc/api.cpp)
OPTIONS = {"msg_fwd_split": 0, "msg_bwd_split": 1, "msg_fwd_kernel": 2, "grp_waves": 3, "grp_records": 4, "csr_build": 5,
           "pseudo_chunks": 6, "wgrad_tiling": 7, "tile_fwd_lds_min": 8, "bwd_input_waves": 9, "pseudo_fwd": 10, "decoder_fat": 11, "decoder_wlds": 12, "skinny_rows": 13,
           "tile_fwd_bal": 14, "optim_one_launch": 15, "decoder_colsplit": 16, "decoder_nodesplit": 17,
           "msg_fwd_balanced": 18, "bwd_input_split": 19, "msg_bwd_mfma": 20, "streamk": 21}


def set_option(name: str, value: int) -> None:
    """cgv_set_option by name (process wide; set before the launches it should affect)."""
    call("cgv_set_option", OPTIONS[name], int(value))


def get_option(name: str) -> int:
    return int(load().cgv_get_option(OPTIONS[name]))


def reset_options() -> None:
    call("cgv_reset_options")


def header_symbols():
    """Names of all functions declared in include/cgvae_hip.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cgv_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load the library once; raise loudly when it is absent (no CPU / eager fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m coarsegrainingvae_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no fallback path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.cgv_version() < 100:
            raise RuntimeError("libcgvae_hip.so is older than this package")
        _lib = lib
    return _lib


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("coarsegrainingvae_amd kernels need device (hip) tensors; there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError("tensor handed to the C ABI must be contiguous")
    return t.data_ptr()


_SPLIT_WS = {}                       # (device, stream) -> the split reduction's workspace (cgv_tile_bwd_input_split)
_SPLIT_TLS = threading.local()       # the C side keeps the registration per calling thread (autograd runs its own)
# 64 KB of self-resetting tickets + partial tiles: the split reduction of tile_bwd_input_k needs <= 8 MB, the stream-K kernel
# (csrc/streamk_gemm.hip) two 64 KB partial tiles per block of its grid of <= 512
_SPLIT_BYTES = 64 * 1024 + 64 * 1024 * 1024


def prepare_split_workspace(stream=None):
    """Create (zero-filled, once) the split reduction's workspace of ``stream`` (default: the current one).  Called before a
    stream capture for the capture stream: a workspace cannot be created inside one (its zero fill would be a recorded node
    that has not run), and a capture without one runs its backward-input products unsplit."""
    st = torch.cuda.current_stream() if stream is None else stream
    key = (st.device.index if st.device.index is not None else torch.cuda.current_device(), int(st.cuda_stream))
    if key not in _SPLIT_WS:
        if torch.cuda.is_current_stream_capturing():
            return None
        _SPLIT_WS[key] = torch.zeros(_SPLIT_BYTES, dtype=torch.uint8, device=f"cuda:{key[0]}")
    return _SPLIT_WS[key]


def split_workspace_ready() -> bool:
    """True when the next backward-input launch of the tile kernels on the current stream may split its reduction (the
    dispatch between the tile and the row-split kernels looks at this: primitives._LinearFn, ops._dense_bwd_input)."""
    if load().cgv_get_option(OPTIONS["bwd_input_split"]) == 1:
        return False
    key = (torch.cuda.current_device(), int(torch.cuda.current_stream().cuda_stream))
    return key in _SPLIT_WS or not torch.cuda.is_current_stream_capturing()


def _register_split_workspace(lib):
    """The backward-input launches of the tile kernels split few-tile / long-reduction products over several blocks when the
    calling thread has registered a workspace for the launch stream: one zero-filled buffer per (device, stream).  The
    launches of one stream are ordered, so they share it; a graph captured with it must not replay concurrently with other
    work of the stream it was captured on."""
    key = (torch.cuda.current_device(), int(torch.cuda.current_stream().cuda_stream))
    if getattr(_SPLIT_TLS, "cur", None) == key:
        return
    ws = prepare_split_workspace()
    if ws is None:
        lib.cgv_tile_bwd_input_split(None, 0, key[1])
        _SPLIT_TLS.cur = None
        return
    if lib.cgv_tile_bwd_input_split(ws.data_ptr(), ws.numel(), key[1]) != 0:
        raise RuntimeError("cgv_tile_bwd_input_split failed: " + lib.cgv_last_error_string().decode("utf-8", "replace"))
    _SPLIT_TLS.cur = key


def call(name: str, *args, tag=None):
    """Invoke a status-returning entry point; raise RuntimeError on any non-zero code.
    ``tag`` names the launch for the optional HIP-event timer (ktimer.py)."""
    lib = load()
    if name.startswith("cgv_tile_") and ("bwd_input" in name or "linear_fwd" in name):
        _register_split_workspace(lib)
    tok = ktimer.begin(tag) if tag is not None else None
    rc = getattr(lib, name)(*args)
    ktimer.end(tok)
    if rc != 0:
        msg = lib.cgv_last_error_string().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed with code {rc}: {msg}")
