"""ctypes binding of libgrouped_cumprod_hip.so (the C ABI of include/grouped_cumprod_hip.h).

There is NO fallback: if the library is missing or does not load, importing the
ops raises.  The reference has the same property — `import grouped_cumprod`
fails when its extension is not built (reference: gs_model.py:8).
"""
import ctypes
import os

from . import _build
from ._build import LIB_PATH

_c_void_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_sz = ctypes.c_size_t
_i32 = ctypes.c_int32

# name -> (restype, argtypes); must list every symbol include/grouped_cumprod_hip.h declares
SIGNATURES = {
    "gcp_abi_version": (ctypes.c_int, []),
    "gcp_source_hash": (ctypes.c_char_p, []),
    "gcp_last_hip_error": (ctypes.c_int, []),
    "gcp_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "gcp_workspace_bytes": (_sz, [_i64]),
    "gcp_workspace_init": (ctypes.c_int, [_c_void_p, _sz, _c_void_p]),
    "gcp_cumprod_forward": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _sz, _c_void_p]),
    "gcp_cumsum_forward": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _sz, _c_void_p]),
    "gcp_cumsum_reverse": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _sz, _c_void_p]),
    "gcp_cumprod_forward_indexed": (ctypes.c_int, [_c_void_p] * 4 + [_i64, _c_void_p, _sz, _c_void_p]),
    "gcp_cumsum_forward_indexed": (ctypes.c_int, [_c_void_p] * 4 + [_i64, _c_void_p, _sz, _c_void_p]),
    "gcp_cumsum_reverse_indexed": (ctypes.c_int, [_c_void_p] * 4 + [_i64, _c_void_p, _sz, _c_void_p]),
    "gcp_cumprod_forward_carry": (ctypes.c_int, [_c_void_p] * 4 + [_i64, _i64, _c_void_p, _sz, _c_void_p]),
    "gcp_cumsum_forward_carry": (ctypes.c_int, [_c_void_p] * 4 + [_i64, _i64, _c_void_p, _sz, _c_void_p]),
    "gcp_cumsum_reverse_carry": (ctypes.c_int, [_c_void_p] * 4 + [_i64, _i64, _c_void_p, _sz, _c_void_p]),
    "gcp_cumprod_backward": (
        ctypes.c_int,
        [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _i64, _c_void_p, _sz, _c_void_p],
    ),
    "gcp_check_groups": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i64, ctypes.POINTER(_i64), _c_void_p]),
    "gcp_check_permutation": (ctypes.c_int, [_c_void_p, _i64, ctypes.POINTER(_i64), _c_void_p]),
    "gcp_check_group_ids": (ctypes.c_int, [_c_void_p, _i64, _i64, ctypes.POINTER(_i64), _c_void_p]),
    "gcp_set_validate_operands": (ctypes.c_int, [ctypes.c_int]),
    "gcp_tile_elems": (ctypes.c_int, []),
    "gcp_last_fallback_tiles": (ctypes.c_int, [_c_void_p, _c_void_p, ctypes.POINTER(_i64)]),
    "gcp_last_lookback_tiles": (ctypes.c_int, [_c_void_p, _c_void_p, ctypes.POINTER(_i64)]),
    "gcp_set_lookback_wait_us": (ctypes.c_int, [_i64]),
    # rows f1 / f2 (gcp_raster.hip)
    "gcp_tile_grid": (ctypes.c_int, [_i32, _i32, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "gcp_scan_i32_workspace_bytes": (_sz, [_i64]),
    "gcp_exclusive_scan_i32": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _c_void_p, _sz, _c_void_p]),
    "gcp_bin_tiles_count": (
        ctypes.c_int,
        [_c_void_p, _c_void_p, _i64, _i32, _i32, _c_void_p, ctypes.POINTER(_i64), _c_void_p, _sz, _c_void_p],
    ),
    "gcp_bin_workspace_bytes": (_sz, [_i64, _i64]),
    "gcp_bin_tiles_fill": (
        ctypes.c_int,
        [_c_void_p, _c_void_p, _i64, _i32, _i32, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _sz, _c_void_p],
    ),
    "gcp_bin_tiles": (
        ctypes.c_int,
        [_c_void_p, _c_void_p, _i64, _i32, _i32, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _sz, _c_void_p],
    ),
    "gcp_blend_checkpoint_floats": (_sz, [_i64, _i32, _i32]),
    "gcp_blend_forward": (
        ctypes.c_int,
        [_c_void_p] * 6 + [_i64, _i32, _i32, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    ),
    "gcp_blend_backward_workspace_bytes": (_sz, [_i64]),
    "gcp_blend_backward": (
        ctypes.c_int,
        [_c_void_p] * 6 + [_i64, _i32, _i32, _c_void_p, _i64] + [_c_void_p] * 8 + [_c_void_p, _sz, _c_void_p],
    ),
    "gcp_gather_f32": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p]),
    "gcp_unsort_finish": (ctypes.c_int, [_c_void_p] * 5 + [_i64, _i32, _c_void_p]),
    "gcp_sort_workspace_bytes": (_sz, [_i64]),
    "gcp_sort_pairs_u32": (ctypes.c_int, [_c_void_p, _i64, _i32, _c_void_p, _c_void_p, _c_void_p, _sz, _c_void_p]),
    "gcp_sort_rects": (ctypes.c_int, [_c_void_p, _i64, _i32, _i32, _c_void_p, _c_void_p, _c_void_p, _sz, _c_void_p]),
    "gcp_rects_key_range": (ctypes.c_int, [_c_void_p, _i64, _c_void_p, _c_void_p]),
    "gcp_compact_workspace_bytes": (_sz, [_i64]),
    "gcp_compact_finish": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i64, _i32, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _sz, _c_void_p]),
    "gcp_pairs_finish_prepare": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _c_void_p]),
    "gcp_pairs_finish_boxes": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i32, _i32] + [_c_void_p] * 6 + [_i64, _i32, _c_void_p, _i32, _c_void_p]),
    "gcp_compact_kept_workspace_bytes": (_sz, [_i64]),
    "gcp_compact_kept_count": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i64, _i64, _c_void_p, _c_void_p, _sz, _c_void_p]),
    "gcp_compact_kept_write": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i64, _c_void_p, _c_void_p, _sz, _c_void_p]),
    "gcp_rects_cut_workspace_bytes": (_sz, [_i64, _i64, _i64, _i32, _i64]),
    "gcp_rects_cut": (ctypes.c_int, [_c_void_p, _i32, _i64, _i64, _i64, _i32, _i64] + [_c_void_p] * 6 + [_sz, _c_void_p]),
    "gcp_pixels_range": (ctypes.c_int, [_c_void_p, _i32, _i64, _c_void_p, _c_void_p]),
    "gcp_pixels_min_workspace_bytes": (_sz, [_i32, _i32]),
    "gcp_pixels_min": (ctypes.c_int, [_c_void_p, _i32, _c_void_p, _i64, _i32, _i32, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _sz, _c_void_p]),
    "gcp_rects_rows_workspace_bytes": (_sz, [_i64]),
    "gcp_rects_rows_capacity": (_i64, [_i64]),
    "gcp_rects_rows": (ctypes.c_int, [_c_void_p, _i64, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _sz, _c_void_p]),
    "gcp_rects_rows_i64": (ctypes.c_int, [_c_void_p, _i64, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _sz, _c_void_p]),
    "gcp_rows_rectangles_workspace_bytes": (_sz, [_i64]),
    "gcp_rows_rectangles": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _sz, _c_void_p]),
    "gcp_rectangle_boxes": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _i64, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "gcp_pairs_scan_boxes": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i32, _i32] + [_c_void_p] * 5 + [_i64, _i32, _c_void_p, _c_void_p]),
    "gcp_box_sizes": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i32, _i32, _c_void_p, _c_void_p]),
    "gcp_expand_rects": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _i64, _i64, _i32, _i32, _c_void_p, _c_void_p, _c_void_p]),
    "gcp_pixel_lists_count": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i32, _i32] + [_c_void_p] * 5),
    "gcp_pixel_lists_fill": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i32, _i32] + [_c_void_p] * 8),
    "gcp_project_forward": (ctypes.c_int, [_c_void_p] * 7 + [_i64, _i32, _i32, _i32, _i32, ctypes.c_float] + [_c_void_p] * 5),
    "gcp_project_gather": (ctypes.c_int, [_c_void_p, _c_void_p, _i64] + [_c_void_p] * 11),
    "gcp_adam_step": (ctypes.c_int, [_c_void_p] * 4 + [_i64] + [ctypes.c_double] * 4 + [_i64, _c_void_p]),
    "gcp_ssim_blocks": (_i64, [_i64, _i32, _i32]),
    "gcp_ssim_l1_forward": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _i32, _i32, _c_void_p, ctypes.c_float, ctypes.c_float]
                            + [_c_void_p] * 5),
    "gcp_ssim_l1_backward": (ctypes.c_int, [_c_void_p] * 5 + [_i64, _i32, _i32, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "gcp_project_backward": (ctypes.c_int, [_c_void_p] * 7 + [_i64, _i32, _i32] + [_c_void_p] * 10),
}

ABI_VERSION = 4

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("GCP_LIBRARY", LIB_PATH)
    if path == LIB_PATH and os.path.exists(path) and _build.is_stale():
        # a binary older than the sources of this checkout: rebuild where hipcc exists, refuse otherwise
        try:
            _build.build_hip_library(force=True)
        except RuntimeError as e:
            raise ImportError(f"{path} was built from other sources than this checkout's and cannot be rebuilt: {e}") from e
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python setup.py build_ext --inplace` "
            "(or __graft_entry__.build()). There is no CPU fallback."
        )
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.gcp_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {got}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(status, what):
    """Turn a GCP_* status into the RuntimeError the reference's callers would see."""
    if status == 0:
        return
    lib = load()
    msg = lib.gcp_status_string(status).decode()
    if status == 3:
        msg += f" (hipError {lib.gcp_last_hip_error()})"
    raise RuntimeError(f"{what}: {msg}")
