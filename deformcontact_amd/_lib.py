"""ctypes binding of ``libdeformcontact_hip.so`` (the C ABI in ``include/deformcontact.h``).

There is deliberately NO fallback: if the library is missing or fails to load,
every op raises.  The CPU oracle under ``oracle/`` is test infrastructure and is
never imported from here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libdeformcontact_hip.so")

_lib = None

_vp = c_void_p
_SIGNATURES = {
    "dc_version": (c_int, []),
    "dc_last_error": (c_char_p, []),
    "dc_stream_capture_id": (c_int64, [_vp]),
    "dc_csr_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "dc_csr_build": (c_int, [_vp, c_int64, c_int64, c_int, c_int, _vp, _vp, _vp, _vp, _vp, _vp,
                             _vp, c_int64, _vp]),
    "dc_graph_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "dc_graph_build": (c_int, [_vp, c_int64, c_int64, c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, c_int64, _vp]),
    "dc_graph_build_parts": (c_int, [POINTER(_vp), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), c_int,
                                     c_int64, c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, c_int64,
                                     _vp]),
    "dc_graph_build_plan": (c_int, [c_int64, c_int64, c_int, POINTER(c_int), POINTER(c_int)]),
    "dc_graph_build_segmented": (c_int, [_vp, c_int64, c_int64, POINTER(c_int64), POINTER(c_int64), c_int,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dc_attn_flash_prep": (c_int, [_vp, c_int64, c_int64, _vp, _vp, _vp]),
    "dc_attn_flash_ds": (c_int, [_vp, c_int64, _vp, _vp, c_int64, _vp, _vp, _vp, _vp, _vp, _vp, c_int64, c_int64,
                                 c_int64, c_int64, _vp, _vp, c_int64, _vp, _vp, _vp, _vp]),
    "dc_tag_linear_fwd_h2p_corr": (c_int, [_vp, c_int64, _vp, _vp, _vp, _vp, c_int64, c_int64, c_int64, c_int64, _vp, _vp,
                                           _vp]),
    "dc_tag_linear_bwd_dw_h2_corr": (c_int, [_vp, c_int64, _vp, _vp, POINTER(_vp), POINTER(c_int64), c_int, POINTER(_vp),
                                             c_int, c_int64, c_int, _vp, c_int64, c_int64, c_int64, c_int64, _vp, _vp,
                                             _vp]),
    "dc_attn_flash_fwd": (c_int, [_vp, c_int64, _vp, _vp, _vp, _vp, _vp, c_int64, c_int64, c_int64, c_int64, _vp,
                                  c_int64, _vp, _vp]),
    "dc_hop_chain_max_nodes": (c_int64, []),
    "dc_hop_chain_lds_request": (c_int64, []),
    "dc_hop_chain_f32": (c_int, [_vp, _vp, _vp, _vp, c_int64, POINTER(c_int64), c_int, _vp, c_int64, c_int64, c_int64,
                                 c_int, c_int, c_int, _vp, c_int, _vp]),
    "dc_spmm_f32_bias_act": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp, c_int, _vp, c_int64, c_int64, c_int64, _vp]),
    "dc_colsum_workspace_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "dc_mask_colsum_f32": (c_int, [_vp, c_int64, _vp, c_int64, _vp, c_int64, c_int64, c_int64, _vp, c_int64, _vp, c_int,
                                   _vp]),
    "dc_gat_alpha_fwd": (c_int, [_vp, c_int64, _vp, _vp, _vp, _vp, c_int64, c_int64, _vp]),
    "dc_gat_alpha_bwd": (c_int, [_vp, c_int64, _vp, _vp, _vp, _vp, _vp, c_int64, c_int64, c_int64, _vp, c_int64, _vp, _vp,
                                 c_int, _vp]),
    "dc_spmm_f32_pack": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp, c_int64, c_int64, c_int64, c_int64, c_int64, _vp]),
    "dc_spmm_f32_window": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp, c_int64, _vp, c_int64, c_int64,
                                   c_int64, c_int64, _vp]),
    "dc_spmm_f32_rowmax_window": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp, c_int64, _vp, c_int64, c_int64,
                                          c_int64, _vp, c_int, c_int64, _vp]),
    "dc_tag_grouped_weight_prep": (c_int, [POINTER(_vp), c_int, c_int, c_int64, c_int64, POINTER(_vp),
                                           POINTER(_vp), POINTER(_vp), POINTER(_vp), _vp]),
    "dc_tag_grouped_fwd_h2p": (c_int, [_vp, c_int64, c_int, POINTER(c_int64), POINTER(c_int64), c_int64,
                                       POINTER(_vp), POINTER(_vp), c_int, _vp, c_int64, c_int64, c_int64, _vp,
                                       POINTER(_vp), _vp]),
    "dc_tag_grouped_mask_grad": (c_int, [POINTER(_vp), POINTER(c_int64), c_int, POINTER(c_int64),
                                         POINTER(c_int64), c_int64, _vp, c_int64, _vp, c_int64, c_int64, _vp,
                                         _vp, _vp]),
    "dc_tag_grouped_bwd_dw_workspace_bytes": (c_int64, [POINTER(c_int64), c_int, c_int64, c_int64, c_int]),
    "dc_tag_grouped_bwd_dw_h2": (c_int, [_vp, c_int64, POINTER(_vp), POINTER(c_int64), c_int, c_int,
                                         POINTER(c_int64), POINTER(c_int64), c_int64, POINTER(_vp),
                                         POINTER(_vp), c_int, _vp, c_int64, c_int64, c_int64, _vp, _vp, _vp]),
    "dc_generic_dense_launches": (c_int64, [c_int]),
    "dc_kernel_trace": (None, [c_int]),
    "dc_kernel_trace_dump": (c_int64, [c_char_p, c_int64]),
    "dc_hash_i64": (c_int, [_vp, c_int64, _vp, _vp]),
    "dc_morton_codes": (c_int, [_vp, c_int64, c_int64, POINTER(c_float), POINTER(c_float), _vp, _vp]),
    "dc_morton_order_workspace_bytes": (c_int64, [c_int64]),
    "dc_morton_order": (c_int, [_vp, c_int64, c_int64, _vp, _vp, _vp, c_int64, _vp]),
    "dc_relabel_edges": (c_int, [_vp, c_int64, _vp, c_int64, _vp, _vp]),
    "dc_gather_rows": (c_int, [_vp, c_int64, _vp, _vp, c_int64, c_int64, c_int64, _vp]),
    "dc_invert_perm": (c_int, [_vp, _vp, _vp, c_int64, _vp]),
    "dc_spmm_f32": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp, c_int64, _vp, c_int64, c_int64,
                            c_int64, _vp]),
    "dc_spmm_bf16": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp, c_int64, _vp, c_int64, c_int64,
                             c_int64, c_int, _vp]),
    "dc_tag_linear_fwd_bf16": (c_int, [_vp, c_int64, _vp, _vp, c_int, _vp, c_int64, c_int, c_int64, c_int64,
                                       c_int64, _vp]),
    "dc_tag_mask_grad_bf16": (c_int, [_vp, c_int64, c_int, _vp, c_int64, c_int, _vp, c_int64, c_int64, c_int64, _vp]),
    "dc_tag_linear_bwd_dw_bf16_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64, c_int]),
    "dc_tag_linear_bwd_dw_bf16": (c_int, [_vp, c_int64, _vp, c_int64, c_int, POINTER(_vp), _vp, c_int, _vp, c_int64,
                                          c_int64, c_int64, c_int64, _vp]),
    "dc_to_bf16": (c_int, [POINTER(_vp), c_int, c_int64, c_int64, c_int64, _vp, c_int64, _vp]),
    "dc_contact_loss_workspace_bytes": (c_int64, [c_int64]),
    "dc_contact_loss": (c_int, [_vp, _vp, _vp, _vp, _vp, c_int64, _vp, c_int64, c_int64, c_int64, _vp, _vp,
                                _vp, _vp, c_int64, _vp]),
    "dc_tag_linear_fwd": (c_int, [POINTER(_vp), POINTER(c_int64), POINTER(_vp), c_int, _vp, c_int,
                                  _vp, c_int64, c_int64, c_int64, c_int64, _vp]),
    "dc_tag_linear_bwd_dx": (c_int, [_vp, c_int64, _vp, c_int64, POINTER(_vp), c_int,
                                     POINTER(_vp), POINTER(c_int64), c_int64, c_int64, c_int64,
                                     _vp]),
    "dc_tag_linear_fwd_split": (c_int, [POINTER(_vp), POINTER(c_int64), POINTER(_vp), c_int, _vp,
                                        c_int, _vp, c_int64, c_int64, c_int64, c_int64, c_int, _vp]),
    "dc_tag_linear_bwd_dx_split_workspace_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "dc_tag_linear_bwd_dx_split": (c_int, [_vp, c_int64, _vp, c_int64, POINTER(_vp), c_int,
                                           POINTER(_vp), POINTER(c_int64), _vp, c_int64, c_int64,
                                           c_int64, c_int64, c_int, _vp]),
    "dc_tag_linear_bwd_dw_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64, c_int]),
    "dc_tag_linear_bwd_dw": (c_int, [_vp, c_int64, _vp, c_int64, POINTER(_vp), POINTER(c_int64),
                                     c_int, POINTER(_vp), c_int, c_int64, _vp, c_int, _vp, c_int64,
                                     c_int64, c_int64, c_int64, _vp]),
    "dc_tag_linear_bwd_dw_split": (c_int, [_vp, c_int64, _vp, c_int64, POINTER(_vp),
                                           POINTER(c_int64), c_int, POINTER(_vp), c_int, c_int64,
                                           _vp, c_int, _vp, c_int64, c_int64, c_int64, c_int64,
                                           c_int, _vp]),
    "dc_tag_linear_fwd_h2": (c_int, [POINTER(_vp), POINTER(c_int64), POINTER(_vp), c_int, _vp,
                                     c_int, _vp, c_int64, c_int64, c_int64, c_int64, _vp, _vp, _vp]),
    "dc_tag_linear_bwd_dx_h2": (c_int, [_vp, c_int64, _vp, c_int64, POINTER(_vp), c_int,
                                        POINTER(_vp), POINTER(c_int64), _vp, c_int64, c_int64,
                                        c_int64, c_int64, _vp, _vp, _vp]),
    "dc_tag_linear_bwd_dw_h2": (c_int, [_vp, c_int64, _vp, c_int64, POINTER(_vp),
                                        POINTER(c_int64), c_int, POINTER(_vp), c_int, c_int64,
                                        _vp, c_int, _vp, c_int64, c_int64, c_int64, c_int64,
                                        _vp, _vp, _vp]),
    "dc_tag_mask_grad": (c_int, [_vp, c_int64, _vp, c_int64, _vp, c_int64, c_int64, c_int64, _vp, _vp, _vp]),
    "dc_tag_transpose_weights": (c_int, [POINTER(_vp), c_int, c_int64, c_int64, _vp, _vp]),
    "dc_tag_linear_fwd_h2p_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "dc_tag_linear_fwd_h2p": (c_int, [_vp, c_int64, _vp, _vp, c_int, _vp, c_int64, c_int64, c_int64, c_int64,
                                      _vp, _vp, _vp, c_int64, _vp]),
    "dc_tag_linear_fwd_h2p_exp": (c_int, [_vp, c_int64, _vp, _vp, c_int64, c_int64, c_int64, c_int64, _vp, _vp, _vp,
                                          c_int64, _vp]),
    "dc_tag_weight_prep": (c_int, [POINTER(_vp), c_int, c_int64, c_int64, _vp, _vp, _vp, _vp, _vp]),
    "dc_tag_weight_prep_zero": (c_int, [POINTER(_vp), c_int, c_int64, c_int64, _vp, _vp, _vp, _vp, _vp, c_int64, _vp]),
    "dc_rowabsmax_f32": (c_int, [_vp, c_int64, c_int64, c_int64, _vp, _vp]),
    "dc_tag_weight_rowmax": (c_int, [POINTER(_vp), c_int, c_int64, c_int64, _vp, _vp]),
    "dc_spmm_f32_rowmax": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp, c_int64, _vp, c_int64, c_int64,
                                   c_int64, _vp, c_int, _vp]),
    "dc_tag_pack_input": (c_int, [_vp, c_int64, _vp, c_int64, c_int64, c_int64, c_int64, c_int64, _vp]),
    "dc_tag_pack_weights": (c_int, [POINTER(_vp), c_int, _vp, c_int64, c_int64, c_int64, _vp]),
    "dc_tag_linear_fwd_narrow_ok": (c_int, [c_int64, c_int, c_int64, c_int64]),
    "dc_tag_linear_fwd_narrow": (c_int, [_vp, c_int64, POINTER(_vp), c_int, c_int64, _vp, c_int, _vp, c_int64, c_int64,
                                         c_int64, c_int64, _vp]),
    "dc_attn_softmax_rows": (c_int, [_vp, c_int64, c_int64, c_int64, c_int64, _vp, _vp]),
    "dc_attn_exp_rows": (c_int, [_vp, c_int64, c_int64, c_int64, c_int64, _vp, _vp]),
    "dc_attn_ds_rows": (c_int, [_vp, _vp, c_int64, c_int64, c_int64, _vp, _vp, _vp]),
    "dc_adam_flat": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp, c_float, c_float, c_float, c_float,
                             c_int, _vp]),
    "dc_compose_perm": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp]),
    "dc_gat_edge_softmax_fwd": (c_int, [_vp, _vp, _vp, _vp, c_float, _vp, c_int64, _vp]),
    "dc_gat_edge_softmax_bwd": (c_int, [_vp, _vp, _vp, _vp, c_float, _vp, _vp, _vp, _vp, c_int64,
                                        _vp]),
    "dc_sddmm_f32": (c_int, [_vp, _vp, _vp, c_int64, _vp, c_int64, _vp, c_int64, c_int64, _vp]),
    "dc_segment_sum_f32": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp]),
    "dc_gather_f32": (c_int, [_vp, _vp, _vp, _vp, c_int64, _vp]),
}


class DeformContactLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise DeformContactLibraryError(
                f"{SO_PATH} not found. Build it with `python -m deformcontact_amd.build` "
                "(hipcc --offload-arch=gfx950). deformcontact_amd has no CPU/PyTorch fallback.")
        try:
            handle = ctypes.CDLL(SO_PATH)
        except OSError as e:  # pragma: no cover
            raise DeformContactLibraryError(f"cannot load {SO_PATH}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise DeformContactLibraryError(
                    f"{SO_PATH} does not export {name}; rebuild the library") from e
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().dc_last_error()
        raise DeformContactLibraryError(
            f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def kernel_trace(on: bool) -> None:
    """Start (clearing the log) / stop the library's launch log (``dc_kernel_trace``)."""
    lib().dc_kernel_trace(1 if on else 0)


def kernel_trace_counts() -> dict:
    """``{kernel name: launches}`` recorded since ``kernel_trace(True)``."""
    n = int(lib().dc_kernel_trace_dump(None, 0))
    buf = ctypes.create_string_buffer(n + 1)
    lib().dc_kernel_trace_dump(buf, n + 1)
    out = {}
    for line in buf.value.decode().splitlines():
        name, _, cnt = line.rpartition(" ")
        out[name] = int(cnt)
    return out


def exported_names():
    return list(_SIGNATURES)
