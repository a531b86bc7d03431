"""ctypes loader for the in-tree HIP engine (gffx_amd/lib/libgffx_hip.so, C-ABI include/gffx_hip.h).

The library is the product's only compute path: loading fails loudly when it has not been
built, and every compute call fails (GFFX_E_NO_DEVICE) when no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgffx_hip.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p

# every symbol include/gffx_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "gffx_hip_abi_version": (C.c_int, []),
    "gffx_hip_device_count": (C.c_int, []),
    "gffx_hip_warmup": (C.c_int, [C.c_int]),
    "gffx_hip_last_error": (C.c_char_p, []),
    "gffx_hip_index_create": (C.c_int, [C.c_uint32, u32p, u32p, u32p, u32p, C.c_int, C.POINTER(vp)]),
    "gffx_hip_index_destroy": (None, [vp]),
    "gffx_hip_index_n_chr": (C.c_uint32, [vp]),
    "gffx_hip_index_n_roots": (C.c_uint64, [vp]),
    "gffx_hip_index_device": (C.c_int, [vp]),
    "gffx_hip_index_sorted_fids": (u32p, [vp]),
    "gffx_hip_index_clone": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "gffx_hip_regions_create": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "gffx_hip_regions_destroy": (None, [vp]),
    "gffx_hip_regions_staging": (vp, [vp, C.c_int]),
    "gffx_hip_regions_wait_staging": (C.c_int, [vp, C.c_int]),
    "gffx_hip_regions_append": (C.c_int, [vp, C.c_int, C.c_uint64]),
    "gffx_hip_regions_append_parts": (C.c_int, [vp, C.c_int, C.c_uint32, u64p, u64p]),
    "gffx_hip_regions_rows": (C.c_uint64, [vp]),
    "gffx_hip_batch_set_regions_store": (C.c_int, [vp, vp, C.c_int, C.c_uint64, C.c_uint64]),
    "gffx_hip_allgather_counts": (C.c_int, [C.c_int, C.POINTER(C.c_int), u64p, u64p]),
    "gffx_hip_lines_test_store": (C.c_int, [vp, vp, C.c_uint32, C.c_int, u8p]),
    "gffx_hip_batch_create": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "gffx_hip_batch_destroy": (None, [vp]),
    "gffx_hip_batch_set_regions_host": (C.c_int, [vp, u32p, C.c_uint64]),
    "gffx_hip_batch_set_regions_soa_host": (C.c_int, [vp, u32p, u32p, u32p, C.c_uint64]),
    "gffx_hip_batch_set_regions_device": (C.c_int, [vp, vp, vp, vp, C.c_uint64]),
    "gffx_hip_batch_run": (C.c_int, [vp, C.c_int, C.c_int, C.c_uint32, C.c_int]),
    "gffx_hip_batch_wait": (C.c_int, [vp]),
    "gffx_hip_batch_sync": (C.c_int, [vp]),
    "gffx_hip_batch_n_queries": (C.c_uint64, [vp]),
    "gffx_hip_batch_total_hits": (C.c_uint64, [vp]),
    "gffx_hip_batch_copy_counts": (C.c_int, [vp, u32p]),
    "gffx_hip_batch_copy_offsets": (C.c_int, [vp, u64p]),
    "gffx_hip_batch_copy_offsets32": (C.c_int, [vp, u32p]),
    "gffx_hip_batch_copy_segbase": (C.c_int, [vp, u64p]),
    "gffx_hip_batch_copy_query_records": (C.c_int, [vp, u32p, u32p, u64p]),
    "gffx_hip_batch_copy_fids": (C.c_int, [vp, u32p]),
    "gffx_hip_batch_copy_triples": (C.c_int, [vp, u32p]),
    "gffx_hip_batch_copy_root_bitmap": (C.c_int, [vp, u64p, C.c_uint64]),
    "gffx_hip_batch_device_counts": (vp, [vp]),
    "gffx_hip_batch_device_fids": (vp, [vp]),
    "gffx_hip_batch_device_triples": (vp, [vp]),
    "gffx_hip_batch_device_regions": (vp, [vp]),
    "gffx_hip_batch_device_offsets": (vp, [vp]),
    "gffx_hip_batch_device_offsets32": (vp, [vp]),
    "gffx_hip_batch_device_segbase": (vp, [vp]),
    "gffx_hip_batch_reserve_hits": (C.c_int, [vp, C.c_uint64]),
    "gffx_hip_batch_kept_pairs_accumulated": (C.c_int, [vp, u64p]),
    "gffx_hip_batch_set_option": (C.c_int, [vp, C.c_char_p, C.c_long]),
    "gffx_hip_batch_options": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "gffx_hip_index_options": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "gffx_hip_batch_set_profiling": (C.c_int, [vp, C.c_int]),
    "gffx_hip_batch_kernel_ms": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), u64p]),
    "gffx_hip_batch_reset_profile": (C.c_int, [vp]),
    "gffx_hip_batch_block_threads": (C.c_uint32, [vp]),
    "gffx_hip_batch_block_count": (C.c_uint32, [vp]),
    "gffx_hip_batch_wide_form": (C.c_int, [vp]),
    "gffx_hip_batches_run_n": (C.c_int, [C.POINTER(vp), C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint64]),
    "gffx_hip_batches_plan": (C.c_int, [C.POINTER(vp), C.c_uint32, u32p, u32p, u32p]),
    "gffx_hip_batches_timed_runs": (C.c_int, [C.POINTER(vp), C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32,
                                    C.POINTER(C.c_double), u32p]),
    "gffx_hip_batch_timed_runs": (C.c_int, [vp, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_double)]),
    "gffx_hip_query_features": (C.c_int, [vp, u32p, C.c_uint64, C.c_int, C.c_int, C.POINTER(u32p), u64p]),
    "gffx_hip_free_host": (None, [vp]),
    "gffx_hip_lines_create": (C.c_int, [C.c_int, C.c_uint64, u32p, u32p, u32p, C.POINTER(vp)]),
    "gffx_hip_lines_destroy": (None, [vp]),
    "gffx_hip_lines_test": (C.c_int, [vp, u32p, C.c_uint64, C.c_uint32, C.c_int, u8p]),
    "gffx_hip_lines_test_device": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.c_int, u8p]),
    "gffx_hip_lines_last_kernel_ms": (C.c_double, [vp]),
    "gffx_hip_lines_last_prep_ms": (C.c_double, [vp]),
    "gffx_hip_lines_last_sort_passes": (C.c_int, [vp]),
    "gffx_hip_lines_copy_tables": (C.c_int, [vp, u64p, u32p, u32p, u32p, u32p]),
    "gffx_hip_lines_copy_dirs": (C.c_int, [vp, u64p, u32p, u32p]),
    "gffx_hip_lines_copy_degenerate": (C.c_int, [vp, u64p, u64p, u32p]),
    "gffx_hip_segments_covered": (C.c_int, [C.c_int, C.c_uint64, u32p, u32p, u32p, u32p, C.c_uint64, C.c_uint32, u32p]),
    "gffx_hip_depth_create": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, u64p, u32p, u32p, u32p, C.c_uint32, u32p,
                                        C.POINTER(vp)]),
    "gffx_hip_depth_destroy": (None, [vp]),
    "gffx_hip_depth_accumulate": (C.c_int, [vp, vp]),
    "gffx_hip_depth_reset": (C.c_int, [vp]),
    "gffx_hip_depth_copy": (C.c_int, [vp, u64p, u32p, u32p]),
}

_lib = None


class GffxHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("gffx_hip error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load libgffx_hip.so (no fallback: raises if it was not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C gffx_amd/csrc). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError == missing export
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise GffxHipError(rc, lib().gffx_hip_last_error().decode(errors="replace"))
