"""Process-group helper for the one-process-per-GPU launch (`python -m torch.distributed.run ... bench.py`).

The multi-GPU logic itself lives behind the C ABI (csrc/shard_group.hip): range shards, the packed 12-byte (u64 id,
f32 score) record, ONE ncclAllGather per query batch and the merge kernel.  What is left here is what a launcher owns:
handing rank 0's RCCL id to every rank over the existing torch.distributed group, and the query split of replica mode.
"""
from __future__ import annotations

import numpy as np

from .index import COMM_ID_BYTES, HnswIndex, comm_unique_id

# the record that travels in the all-gather (csrc/vdb_shard_wire.hpp, shard_group.hip pack_shard_records): id low, id high, score bits
RECORD_DTYPE = np.dtype([("id_lo", "<u4"), ("id_hi", "<u4"), ("score_bits", "<u4")])
REC_EMPTY = 0xFFFFFFFF  # sentinel of a slot past the shard's result count (with id = ~0)
REC_OVERFLOW = 0xFFFFFFFE  # first record of a query whose count is 0xFFFFFFFF (csrc/vdb_shard_wire.hpp)


def query_slice(nq: int, rank: int, world: int):
    """Replica mode (graph path): rank r searches queries [lo, hi) of every batch; no collective."""
    base, rem = divmod(nq, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(ids: np.ndarray, scores: np.ndarray, counts: np.ndarray) -> np.ndarray:
    """Host restatement of pack_shard_records (the wire format), for launchers and the CPU tests."""
    nq, k = ids.shape
    rec = np.empty((nq, k), dtype=RECORD_DTYPE)
    c = counts.astype(np.int64)
    overflow = c == 0xFFFFFFFF   # a device-resident call whose traversal list overflowed: no records, the marker in slot 0
    live = (np.arange(k)[None, :] < c[:, None]) & ~overflow[:, None]
    u = ids.astype(np.uint64)
    rec["id_lo"] = np.where(live, (u & np.uint64(0xFFFFFFFF)).astype(np.uint32), np.uint32(0xFFFFFFFF))
    rec["id_hi"] = np.where(live, (u >> np.uint64(32)).astype(np.uint32), np.uint32(0xFFFFFFFF))
    rec["score_bits"] = np.where(live, np.ascontiguousarray(scores, dtype=np.float32).view(np.uint32), np.uint32(REC_EMPTY))
    rec["score_bits"][overflow, 0] = np.uint32(REC_OVERFLOW)
    return rec


def join_process_group(index: HnswIndex, rank: int, world: int, device=None) -> None:
    """Every rank calls this once after torch.distributed.init_process_group: rank 0 generates the RCCL id, the torch
    group broadcasts its 128 bytes, every rank joins (vdb_hip_index_join_group)."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros(COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8).clone()
    if dist.is_initialized() and world > 1:
        if device is not None:
            buf = buf.to(device)
        dist.broadcast(buf, src=0)
    index.join_group(bytes(buf.cpu().numpy().tobytes()), rank, world)
