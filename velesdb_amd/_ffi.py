"""ctypes binding of libvelesdb_hip.so — exactly the symbols include/velesdb_hip.h declares.

There is no fallback: if the shared library is missing this module raises at import of the
first symbol, and every compute call fails with VDB_ERR_NO_DEVICE when no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# the in-tree library.  The package reads no environment variable; a test harness or a probe script that wants another build of
# the same ABI (the probe build with its environment switches, a kernel variant) says so in code, BEFORE the first call: use_library()
LIB_PATH = os.path.join(_HERE, "lib", "libvelesdb_hip.so")
PROBE_LIB_PATH = os.path.join(_HERE, "lib", "libvelesdb_hip_probe.so")

VDB_OK = 0
VDB_DUPLICATE_IGNORED = 1
VDB_ERR_INVALID_ARG = -1
VDB_ERR_DIM_MISMATCH = -2
VDB_ERR_NO_DEVICE = -3
VDB_ERR_HIP = -4
VDB_ERR_IO = -5
VDB_ERR_OOM = -6
VDB_ERR_UNSUPPORTED = -7
VDB_ERR_STATE = -8

# every function include/velesdb_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _u32, _u64, _f32 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_float
_pi32, _pu32, _pu64, _pf32 = C.POINTER(_i32), C.POINTER(_u32), C.POINTER(_u64), C.POINTER(_f32)
SIGNATURES = {
    "vdb_hip_device_count": (_i32, [_pi32]),
    "vdb_hip_device_name": (_i32, [_i32, C.c_char_p, C.c_size_t]),
    "vdb_hip_index_create": (_i32, [_u32, _i32, _u32, _u32, _u64, _pi32, _i32, _i32, C.POINTER(_vp)]),
    "vdb_hip_comm_unique_id": (_i32, [_vp]),
    "vdb_hip_index_join_group": (_i32, [_vp, _vp, _i32, _i32]),
    "vdb_hip_index_shard_info": (_i32, [_vp, _pi32, _pi32, _pi32, _pi32, _pi32]),
    "vdb_hip_index_destroy": (None, [_vp]),
    "vdb_hip_batch_norm": (_i32, [_i32, _vp, _u64, _u32, _vp]),
    "vdb_hip_normalize_rows": (_i32, [_i32, _vp, _u64, _u32]),
    "vdb_hip_batch_dot_product": (_i32, [_i32, _vp, _u32, _vp, _u64, _u32, _vp]),
    "vdb_hip_batch_hamming_binary": (_i32, [_i32, _vp, _vp, _u64, _u32, _vp]),
    "vdb_hip_batch_jaccard_binary": (_i32, [_i32, _vp, _vp, _u64, _u32, _vp]),
    "vdb_hip_index_tombstone_count": (_i32, [_vp, _pu64]),
    "vdb_hip_index_vacuum": (_i32, [_vp, _pu64]),
    "vdb_hip_index_save_dir": (_i32, [_vp, C.c_char_p]),
    "vdb_hip_index_load_dir": (_i32, [C.c_char_p, _i32, C.POINTER(_vp)]),
    "vdb_hip_index_upload_vector_store": (_i32, [_vp, C.c_char_p, _pu64]),
    "vdb_hip_index_insert": (_i32, [_vp, _u64, _vp, _u32]),
    "vdb_hip_index_insert_batch": (_i32, [_vp, _vp, _vp, _u64, _pu64]),
    "vdb_hip_index_insert_batch_parallel": (_i32, [_vp, _vp, _vp, _u64, _u32, _pu64]),
    "vdb_hip_index_build_graph": (_i32, [_vp, _u32]),
    "vdb_hip_index_enable_bf16": (_i32, [_vp]),
    "vdb_hip_index_set_storage_mode": (_i32, [_vp, _i32]),
    "vdb_hip_index_get_quantized": (_i32, [_vp, _u64, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "vdb_hip_index_train_quantizer": (_i32, [_vp, _u32]),
    "vdb_hip_index_quantizer_trained": (_i32, [_vp, _pi32]),
    "vdb_hip_index_search_with_config": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _i32, _u64, _vp, _vp, _vp]),
    "vdb_hip_set_int8_oversampling": (_i32, [_u32]),
    "vdb_hip_index_upload": (_i32, [_vp, _vp, _vp, _u64, _pu64]),
    "vdb_hip_index_upload_dev": (_i32, [_vp, _u64, _vp, _u64, _vp]),
    "vdb_hip_index_remove": (_i32, [_vp, _u64, _pi32]),
    "vdb_hip_index_len": (_i32, [_vp, _pu64]),
    "vdb_hip_index_dimension": (_i32, [_vp, _pu32]),
    "vdb_hip_index_metric": (_i32, [_vp, _pi32]),
    "vdb_hip_index_node_count": (_i32, [_vp, _pu64]),
    "vdb_hip_index_search": (_i32, [_vp, _vp, _u32, _u32, _u32, _i32, _vp, _vp, _pu32]),
    "vdb_hip_index_search_batch": (_i32, [_vp, _vp, _u32, _u32, _u32, _i32, _vp, _vp, _vp]),
    "vdb_hip_index_search_multi_entry": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "vdb_hip_index_search_rerank": (_i32, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "vdb_hip_index_search_batch_dev": (_i32, [_vp, _vp, _u32, _u32, _u32, _i32, _vp, _vp, _vp, _vp]),
    "vdb_hip_batch_distance": (_i32, [_i32, _i32, _i32, _vp, _vp, _u64, _u32, _vp]),
    "vdb_hip_batch_distance_dev": (_i32, [_i32, _i32, _vp, _vp, _u64, _u32, _vp, _vp]),
    "vdb_hip_index_load_reference_files": (_i32, [_vp, C.c_char_p, C.c_char_p]),
    "vdb_hip_index_save_reference_files": (_i32, [_vp, C.c_char_p, C.c_char_p]),
    "vdb_hip_index_get_neighbors": (_i32, [_vp, _u32, _u64, _vp, _u32, _pu32]),
    "vdb_hip_index_graph_info": (_i32, [_vp, _pu32, _pu32, C.POINTER(C.c_int64)]),
    "vdb_hip_index_last_search_stats": (_i32, [_vp, _pu64, _pu64]),
    "vdb_hip_index_last_prefetch_hits": (_i32, [_vp, _pu64]),
    "vdb_hip_set_kernel_timing": (_i32, [_i32]),
    "vdb_hip_set_max_query_tile": (_i32, [_u32]),
    "vdb_hip_set_sweep_engine": (_i32, [_i32]),
    "vdb_hip_set_split_selector": (_i32, [_i32]),
    "vdb_hip_index_last_split_stats": (_i32, [_vp, _pu32, _pu32]),
    "vdb_hip_index_last_select_level": (_i32, [_vp, C.POINTER(C.c_int32)]),
    "vdb_hip_index_last_kernels": (_i32, [_vp, _pu32]),
    "vdb_hip_index_last_selection_ms": (_i32, [_vp, _pf32, _pu32]),
    "vdb_hip_index_set_option": (_i32, [_vp, _i32, C.c_int64]),
    "vdb_hip_index_get_option": (_i32, [_vp, _i32, C.POINTER(C.c_int64)]),
    "vdb_hip_index_combine_stats": (_i32, [_vp, _pu64, _pu64, _pu64, _pu64]),
    "vdb_hip_index_build_stats": (_i32, [_vp, _pu64, _pu64, _pu64, _pu64]),
    "vdb_hip_index_sweep_arith_mode": (_i32, [_vp, _u32, _pi32]),
    "vdb_hip_index_last_kernel_ms": (_i32, [_vp, _pf32, _pu32]),
    "vdb_hip_last_error": (C.c_char_p, []),
    "vdb_hip_version": (C.c_char_p, []),
}

_lib = None


class VelesHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[vdb status {code}] {msg}")
        self.code = code


def use_library(path: str) -> None:
    """Binds the package to another build of libvelesdb_hip (tests/conftest.py: the probe build; tools/probes: kernel variants).
    Must be called before anything touched the library."""
    global LIB_PATH
    if _lib is not None and os.path.abspath(path) != os.path.abspath(LIB_PATH):
        raise RuntimeError(f"libvelesdb_hip is already loaded from {LIB_PATH}")
    LIB_PATH = path


def lib() -> C.CDLL:
    """Loads the HIP library.  Raises (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension is the product, there is no fallback. "
                "Build it with `python -m velesdb_amd.build` (needs hipcc).")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def last_error() -> str:
    return lib().vdb_hip_last_error().decode()


def check(rc: int) -> int:
    if rc < 0:
        raise VelesHipError(rc, last_error())
    return rc
