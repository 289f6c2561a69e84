"""Multi-GPU composition of the exact sweep: the corpus is range-sharded, shard g holds rows
[offset_g, offset_g + n_g) and answers every query with its local top-k; ONE all-gather of k (id, score) pairs
per query per rank is the only exchange, followed by a (world*k -> k) merge on every rank.

The functions take torch tensors on whatever device the process group works on (RCCL: GPU tensors; gloo: CPU
tensors, used by the world_size-2 CPU tests).  No kernel of the hot path lives here: the per-shard top-k comes
from libvelesdb_hip.so (HnswIndex.search_batch_dev, MODE_BRUTE); this file is the collective + merge only.
The graph path does not shard: replicas + a split query stream (`query_slice`), no collective."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def query_slice(nq: int, rank: int, world: int) -> Tuple[int, int]:
    """Replica mode: contiguous slice [lo, hi) of a batch of nq queries served by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(nq, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_shard_topk(local_ids: torch.Tensor, local_scores: torch.Tensor, local_counts: Optional[torch.Tensor],
                     shard_offset: int, k: int, higher_is_better: bool,
                     group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """local_ids [nq,k] int64 (shard-local row ids), local_scores [nq,k] f32, local_counts [nq] (valid entries per
    query, None = all k valid).  Returns (global ids [nq,k], scores [nq,k], counts [nq]) identical on every rank:
    best first; equal scores ordered by global row id ascending (the canonical tie order of the single-GPU sweep)."""
    world = dist.get_world_size(group)
    nq = local_ids.shape[0]
    dev = local_ids.device
    gids = (local_ids + shard_offset).contiguous()
    sc = local_scores.contiguous()
    if local_counts is None:
        local_counts = torch.full((nq,), k, dtype=torch.int32, device=dev)
    cnt = local_counts.to(torch.int32).contiguous()
    # outputs are the concatenation along dim 0 (the layout both RCCL and gloo accept)
    all_ids = torch.empty((world * nq, k), dtype=gids.dtype, device=dev)
    all_sc = torch.empty((world * nq, k), dtype=sc.dtype, device=dev)
    all_cnt = torch.empty((world * nq,), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_ids, gids, group=group)
    dist.all_gather_into_tensor(all_sc, sc, group=group)
    dist.all_gather_into_tensor(all_cnt, cnt, group=group)
    all_ids = all_ids.view(world, nq, k)
    all_sc = all_sc.view(world, nq, k)
    all_cnt = all_cnt.view(world, nq)
    cand_id = all_ids.permute(1, 0, 2).reshape(nq, world * k)
    cand_sc = all_sc.permute(1, 0, 2).reshape(nq, world * k)
    slot = torch.arange(k, device=dev).view(1, 1, k).expand(world, nq, k)
    valid = (slot < all_cnt.unsqueeze(-1)).permute(1, 0, 2).reshape(nq, world * k)
    # total order key: score (direction by metric), then global id ascending; invalid slots last
    worst = float("-inf") if higher_is_better else float("inf")
    key_sc = torch.where(valid, cand_sc, torch.full_like(cand_sc, worst))
    key_id = torch.where(valid, cand_id, torch.full_like(cand_id, torch.iinfo(cand_id.dtype).max))
    order = torch.argsort(key_id, dim=1, stable=True)                       # secondary key first
    key_sc = torch.gather(key_sc, 1, order)
    order2 = torch.argsort(key_sc, dim=1, descending=higher_is_better, stable=True)
    order = torch.gather(order, 1, order2)[:, :k]
    out_ids = torch.gather(cand_id, 1, order)
    out_sc = torch.gather(cand_sc, 1, order)
    out_cnt = torch.clamp(valid.sum(dim=1), max=k).to(torch.int32)
    return out_ids, out_sc, out_cnt
