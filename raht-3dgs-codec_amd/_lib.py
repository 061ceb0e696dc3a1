"""ctypes loader for libraht_hip.so (the C ABI declared in include/raht.h).

The product path fails loudly when the HIP library is missing -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
SO_PATH = os.path.join(_PKG, "libraht_hip.so")

RAHT_OK = 0
ERRORS = {-1: "RAHT_ERR_INVALID", -2: "RAHT_ERR_UNSORTED", -3: "RAHT_ERR_BOUNDS", -4: "RAHT_ERR_HIP",
          -5: "RAHT_ERR_NOMEM", -6: "RAHT_ERR_UNSUPPORTED"}
RAHT_F32, RAHT_F64, RAHT_I32, RAHT_I64 = 0, 1, 2, 3
ENGINE_TILE, ENGINE_LEVEL = 0, 1

# every symbol include/raht.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "raht_last_error", "raht_version", "raht_plan_create", "raht_plan_create_from_keys", "raht_plan_create_from_keys_borrowed",
    "raht_plan_destroy", "raht_plan_set_top_level", "raht_plan_roots", "raht_plan_set_root_buffer", "raht_plan_size", "raht_plan_nbits", "raht_plan_set_engine", "raht_plan_set_tail_tile", "raht_release_cached_memory", "raht_quant_rows", "raht_dequant_rows",
    "raht_plan_levels", "raht_plan_export_level", "raht_plan_order", "raht_plan_arrays",
    "raht_plan_copy_array", "raht_plan_stage_stats", "raht_fwd", "raht_fwd_f64", "raht_inv", "raht_inv_f64", "raht_debug_run_stage", "raht_plan_set_stage0_events", "raht_fwd_quant", "raht_dequant_inv", "raht_plan_prepare",
    "raht_quant_reorder", "raht_dequant_unreorder", "raht_voxelize", "raht_voxelize_all", "raht_voxelize_plan", "raht_morton", "raht_sort_keys",
    "raht_voxel_keys", "raht_sort_fallbacks", "raht_voxelize_residuals", "raht_plan_set_row_map", "raht_rows_gather", "raht_rows_scatter",
    "raht_plan_set_max_stages", "raht_plan_set_concurrent_directions", "raht_quant_reorder_f64", "raht_dequant_unreorder_f64", "raht_fwd_quant_f64", "raht_dequant_inv_f64",
    "raht_fwd_quant_mixed", "raht_dequant_inv_mixed", "raht_dequant_inv_sqdiff", "raht_fwd_quant_multi", "raht_plan_mixed_stats",
    "raht_fwd_batch", "raht_inv_batch", "raht_fwd_quant_batch", "raht_dequant_inv_batch",
    "raht_rlgr_bound", "raht_rlgr_encode", "raht_rlgr_decode", "raht_rlgr_encode_channels", "raht_rlgr_decode_channels", "raht_transpose_i32", "raht_i32_equal", "raht_sqdiff_columns", "raht_merge_clusters", "raht_voxelize_merge", "raht_rlgr_seg_encode", "raht_rlgr_seg_decode", "raht_rlgr_seg_encode_strided", "raht_rlgr_seg_decode_strided", "raht_rlgr_seg_encode_batch", "raht_rlgr_seg_decode_batch", "raht_rlgr_seg_decode_batch_check", "raht_debug_rlgr_decode_out", "raht_debug_rlgr_encode_out",
    "raht_xchg_bytes", "raht_xchg_alloc", "raht_xchg_open", "raht_xchg_close", "raht_xchg_free", "raht_xchg_gather", "raht_xchg_buffer", "raht_xchg_status",
]


class RahtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


def build(verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", _CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
        print(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("building libraht_hip.so failed")
    return SO_PATH


_lib = None


def lib():
    """Load libraht_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"raht-3dgs-codec_amd: HIP library {SO_PATH} is missing. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C {_CSRC}`). "
            "There is no CPU fallback for the RAHT hot path.")
    try:
        import torch  # noqa: F401  -- make torch's libamdhip64 the process-wide HIP runtime first
    except Exception:
        pass
    L = C.CDLL(SO_PATH)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
    L.raht_last_error.restype = C.c_char_p
    L.raht_plan_create.argtypes = [vp, i32, i64, C.POINTER(dbl), dbl, i32, vp, C.POINTER(vp)]
    L.raht_plan_create_from_keys.argtypes = [vp, i64, i32, vp, vp, C.POINTER(vp)]
    L.raht_plan_create_from_keys_borrowed.argtypes = [vp, i64, i32, vp, vp, C.POINTER(vp)]
    L.raht_plan_destroy.argtypes = [vp]
    L.raht_plan_set_top_level.argtypes = [vp, i32, vp]
    L.raht_plan_roots.argtypes = [vp, C.POINTER(i64), vp, vp]
    L.raht_plan_set_root_buffer.argtypes = [vp, vp]
    L.raht_plan_size.argtypes = [vp]
    L.raht_plan_size.restype = i64
    L.raht_plan_nbits.argtypes = [vp]
    L.raht_plan_set_engine.argtypes = [vp, i32, i32]
    L.raht_plan_set_tail_tile.argtypes = [vp, i32, i32, i32]
    L.raht_plan_levels.argtypes = [vp]
    L.raht_plan_export_level.argtypes = [vp, i32, vp, vp, vp, C.POINTER(i64)]
    L.raht_plan_order.argtypes = [vp, vp, vp]
    L.raht_plan_arrays.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.raht_plan_copy_array.argtypes = [vp, i32, vp, vp]
    L.raht_plan_stage_stats.argtypes = [vp, i32, i32, C.POINTER(i32), C.POINTER(i64), i32, C.POINTER(i32)]
    for f in (L.raht_fwd, L.raht_fwd_f64):
        f.argtypes = [vp, vp, i64, i32, vp, i64, vp, vp]
    for f in (L.raht_inv, L.raht_inv_f64):
        f.argtypes = [vp, vp, i64, i32, vp, i64, vp]
    L.raht_plan_set_stage0_events.argtypes = [vp, vp, vp]
    L.raht_debug_run_stage.argtypes = [vp, i32, i32, vp, i64, i32, vp, i64, vp, i64, C.c_float, i32, vp]
    L.raht_fwd_quant.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_float), i32, vp, i64, vp]
    L.raht_dequant_inv.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_float), i32, vp, i64, vp]
    L.raht_plan_prepare.argtypes = [vp, i32, i32, vp]
    L.raht_fwd_quant_mixed.argtypes = [vp, vp, i64, i32, C.POINTER(dbl), i32, i32, vp, i64, vp]
    L.raht_dequant_inv_mixed.argtypes = [vp, vp, i64, i32, C.POINTER(dbl), i32, i32, vp, i64, vp]
    L.raht_dequant_inv_sqdiff.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_float), i32, vp, i64, vp, i64, vp, vp]
    L.raht_fwd_quant_multi.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_float), i32, C.POINTER(vp), i64, vp]
    L.raht_plan_mixed_stats.argtypes = [vp, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), i32]
    pp, pi64 = C.POINTER(vp), C.POINTER(i64)
    L.raht_fwd_batch.argtypes = [i32, pp, pp, pi64, i32, pp, pi64, vp]
    L.raht_inv_batch.argtypes = [i32, pp, pp, pi64, i32, pp, pi64, vp]
    L.raht_fwd_quant_batch.argtypes = [i32, pp, pp, pi64, i32, C.POINTER(C.c_float), i32, pp, pi64, vp]
    L.raht_dequant_inv_batch.argtypes = [i32, pp, pp, pi64, i32, C.POINTER(C.c_float), i32, pp, pi64, vp]
    L.raht_plan_set_max_stages.argtypes = [vp, i32]
    L.raht_plan_set_concurrent_directions.argtypes = [vp, i32]
    L.raht_plan_set_row_map.argtypes = [vp, vp, i64, vp]
    for f in (L.raht_rows_gather, L.raht_rows_scatter):
        f.argtypes = [vp, i64, vp, i64, i32, i32, vp, i64, vp]
    for f in (L.raht_quant_reorder_f64, L.raht_dequant_unreorder_f64, L.raht_fwd_quant_f64, L.raht_dequant_inv_f64):
        f.argtypes = [vp, vp, i64, i32, C.POINTER(dbl), i32, vp, i64, vp]
    L.raht_quant_reorder.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_float), i32, vp, i64, vp]
    L.raht_dequant_unreorder.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_float), i32, vp, i64, vp]
    L.raht_quant_rows.argtypes = [vp, i64, i64, i32, C.POINTER(C.c_float), i32, vp, vp, i64, vp]
    L.raht_dequant_rows.argtypes = [vp, i64, vp, i64, i32, C.POINTER(C.c_float), i32, vp, i64, vp]
    L.raht_voxelize.argtypes = [vp, i64, i64, i32, C.POINTER(C.c_float), dbl, i32, vp, vp, vp, vp, vp,
                                C.POINTER(i64), C.POINTER(C.c_float), C.POINTER(dbl), C.POINTER(dbl), vp]
    L.raht_sqdiff_columns.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp, vp]
    L.raht_voxelize_all.argtypes = [vp, i64, i64, i32, C.POINTER(C.c_float), dbl, i32, vp, vp, vp, vp, vp, vp, vp,
                                    C.POINTER(i64), C.POINTER(C.c_float), C.POINTER(dbl), C.POINTER(dbl), vp]
    L.raht_voxelize_plan.argtypes = [vp, i64, i64, i32, C.POINTER(C.c_float), dbl, i32, vp, vp, vp, C.POINTER(i64),
                                     C.POINTER(C.c_float), C.POINTER(dbl), C.POINTER(dbl), vp, C.POINTER(vp)]
    L.raht_voxelize_merge.argtypes = [vp, i64, i64, i32, i32, C.POINTER(C.c_float), dbl, i32, vp, vp, vp, vp, vp, C.POINTER(i64),
                                      C.POINTER(C.c_float), C.POINTER(dbl), C.POINTER(dbl), vp]
    L.raht_morton.argtypes = [vp, i64, i32, vp, vp]
    L.raht_voxel_keys.argtypes = [vp, i64, i64, C.POINTER(C.c_float), dbl, i32, vp, vp]
    L.raht_voxelize_residuals.argtypes = [vp, i64, i64, i32, vp, vp, vp, C.POINTER(C.c_float), dbl, vp, vp, vp]
    L.raht_sort_keys.argtypes = [vp, i64, i32, vp, vp, vp]
    L.raht_sort_fallbacks.argtypes = []
    L.raht_sort_fallbacks.restype = i64
    L.raht_rlgr_bound.argtypes = [i64]
    L.raht_rlgr_bound.restype = i64
    L.raht_rlgr_encode.argtypes = [vp, i64, i64, i32, vp, i64, C.POINTER(i64)]
    L.raht_rlgr_decode.argtypes = [vp, i64, i64, i32, vp, i64]
    L.raht_rlgr_encode_channels.argtypes = [vp, i64, i32, i64, i64, i32, vp, i64, vp, i32]
    L.raht_rlgr_decode_channels.argtypes = [vp, i64, vp, i64, i32, i32, vp, i64, i64, i32]
    L.raht_merge_clusters.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.raht_transpose_i32.argtypes = [vp, i64, i64, i64, vp, i64, vp]
    L.raht_i32_equal.argtypes = [vp, vp, i64, i32, C.POINTER(i64)]
    L.raht_rlgr_seg_encode.argtypes = [vp, i64, i32, i64, i32, i32, vp, vp, vp, i64, C.POINTER(i64), vp]
    L.raht_rlgr_seg_decode.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, vp, i64, vp, vp]
    L.raht_rlgr_seg_encode_strided.argtypes = [vp, i64, i32, i64, i64, i32, i32, vp, vp, vp, i64, C.POINTER(i64), vp]
    L.raht_rlgr_seg_decode_strided.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, vp, i64, i64, vp, vp]
    L.raht_debug_rlgr_decode_out.argtypes = [i32]
    L.raht_debug_rlgr_encode_out.argtypes = [i32]
    L.raht_rlgr_seg_encode_batch.argtypes = [i32, C.POINTER(vp), i64, i32, i64, i64, i32, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), vp]
    L.raht_rlgr_seg_decode_batch.argtypes = [i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(vp), i64, i32, i32, i32, C.POINTER(vp), i64, i64, vp, vp]
    L.raht_rlgr_seg_decode_batch_check.argtypes = [i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(vp), i64, i32, i32, i32, C.POINTER(vp), C.POINTER(vp), i64, i64, vp, vp]
    L.raht_xchg_bytes.argtypes = [i32, i64, C.POINTER(i64)]
    L.raht_xchg_alloc.argtypes = [i32, i64, C.POINTER(vp), vp]
    L.raht_xchg_open.argtypes = [vp, C.POINTER(vp)]
    L.raht_xchg_close.argtypes = [vp]
    L.raht_xchg_free.argtypes = [vp]
    L.raht_xchg_gather.argtypes = [vp, i64, C.POINTER(vp), i32, i32, C.c_uint32, vp]
    L.raht_xchg_buffer.argtypes = [vp, i32, i64, C.c_uint32, C.POINTER(vp)]
    L.raht_xchg_status.argtypes = [vp, i32, i64, vp, C.POINTER(i32)]
    _lib = L
    return L


def check(rc):
    if rc != RAHT_OK:
        raise RahtError(rc, lib().raht_last_error().decode("utf-8", "replace"))
