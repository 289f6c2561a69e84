"""velesdb_amd — MI355X-native implementation of velesdb-core's HNSW search hot path.

The product is libvelesdb_hip.so (hand-written HIP for gfx950) behind the C ABI in
include/velesdb_hip.h; this package is the host-side mirror of the reference's
VectorIndex / HnswIndex / DistanceEngine / GpuAccelerator interfaces over that ABI.
"""
from ._ffi import LIB_PATH, VelesHipError, lib  # noqa: F401
from .index import (GpuAccelerator, HipDistance, HnswIndex, NativeHnswIndex, MODE_AUTO, MODE_BRUTE, MODE_BRUTE_BF16, MODE_HNSW, MODE_HNSW_INT8,  # noqa: F401
                    MODE_BRUTE_SQ8, MODE_BRUTE_BINARY, OPT_INT8_OVERSAMPLING, OPT_KERNEL_TIMING, OPT_MAX_QUERY_TILE,
                    OPT_SELECTOR_LEVEL, OPT_SWEEP_ENGINE, OPT_COMBINE_MAX_BATCH, OPT_COMBINE_WINDOW_US, OPT_COMBINE_INFLIGHT, SHARD_RANGE, SHARD_REPLICA, comm_unique_id,
                    device_count, device_name, set_kernel_timing, set_max_query_tile, set_split_selector, set_sweep_engine)
from .index import (KERNEL_BITS, KERNEL_BITS_GEMM, KERNEL_GEMM_BF16, KERNEL_GEMM_BF16_GLDS, KERNEL_GEMM_F32, KERNEL_HNSW, KERNEL_HNSW_INT8,  # noqa: F401
                    KERNEL_SELECT_BF16, KERNEL_SELECT_SPLIT, KERNEL_SQ8, KERNEL_SWEEP_MFMA_BF16, KERNEL_SWEEP_MFMA_F32,
                    KERNEL_SWEEP_VALU)
from .params import DistanceMetric, DualPrecisionConfig, HnswParams, SearchQuality, StorageMode  # noqa: F401

__all__ = ["HnswIndex", "NativeHnswIndex", "HipDistance", "GpuAccelerator", "DistanceMetric", "HnswParams", "SearchQuality", "StorageMode", "DualPrecisionConfig",
           "device_count", "device_name", "comm_unique_id", "SHARD_RANGE", "SHARD_REPLICA", "set_kernel_timing", "set_max_query_tile", "set_sweep_engine", "set_split_selector", "lib", "VelesHipError"]
