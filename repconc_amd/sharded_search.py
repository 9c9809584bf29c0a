"""ADC search over a ROW-SHARDED index: every rank holds rows [offset, offset + n_local) of the
code matrix, searches its shard, and the per-rank top-k lists are all-gathered and merged with the same order
as the single-index search, (score descending, id ascending).  The reference only replicates the index on every
GPU (`co.shard = False`, models/repconc/evaluate_repconc.py:131-134 — also available here: give every rank the
whole index and a slice of the queries); sharding is what makes an index larger than one GPU searchable and
divides the scan time by the number of ranks.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.distributed as dist

from .index import PQIndex


def merge_topk(scores_parts: torch.Tensor, ids_parts: torch.Tensor, k: int):
    """scores_parts / ids_parts: [G, nq, k] per-shard results (ids global, -1 = empty).  Returns the merged
    [nq, k] lists.  Two stable sorts on G*k items per query — bookkeeping, not the scan."""
    G, nq, kk = scores_parts.shape
    s = scores_parts.permute(1, 0, 2).reshape(nq, G * kk)
    i = ids_parts.permute(1, 0, 2).reshape(nq, G * kk)
    big = torch.where(i < 0, torch.full_like(i, torch.iinfo(torch.int64).max), i)
    order = torch.argsort(big, dim=1, stable=True)                 # id ascending
    s, i = torch.gather(s, 1, order), torch.gather(i, 1, order)
    order = torch.argsort(s, dim=1, descending=True, stable=True)  # then score descending, ties keep id order
    return torch.gather(s, 1, order)[:, :k].contiguous(), torch.gather(i, 1, order)[:, :k].contiguous()


def sharded_search(index: PQIndex, q: torch.Tensor, k: int, group=None):
    """`index` holds this rank's rows with `index.id_offset` = global id of its first row."""
    scores, ids = index.search(q, k)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return scores, ids
    from . import ops
    out = merge_topk(ops.all_gather(scores, group), ops.all_gather(ids, group), k)
    ops.comm_check(q.device)                                       # a timed-out exchange is an error, not a result
    return out


def search_virtual_shards(shards: Sequence[PQIndex], q: torch.Tensor, k: int):
    """The same merge with the shards held by one process (single-GPU boxes, tests)."""
    parts = [sh.search(q, k) for sh in shards]
    return merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)


def replicated_search(index, q: torch.Tensor, k: int, *search_args, group=None):
    """The reference's replica mode (`co.shard = False`, evaluate_repconc.py:131-134) in the one-process-per-GPU model,
    and BASELINE configs[3] ("IVF nlist=5000 ADC search at 8 GPUs"): every rank holds the WHOLE index (flat `PQIndex` or
    `IVFPQIndex`, anything with `search(q, k, *search_args)`), takes the contiguous slice [nq r / G, nq (r+1) / G) of the
    query batch, searches it, and one all-gather of the padded (scores, ids) blocks gives every rank the full result in
    query order — identical to the single-index search.  `q` is the same tensor on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return index.search(q, k, *search_args)
    G, r = dist.get_world_size(group), dist.get_rank(group)
    nq = q.shape[0]
    bounds = [(nq * i) // G for i in range(G + 1)]
    most = max(bounds[i + 1] - bounds[i] for i in range(G))
    mine = q[bounds[r]:bounds[r + 1]]
    if mine.shape[0]:
        s, i = index.search(mine, k, *search_args)
    else:
        s = torch.empty((0, k), dtype=torch.float32, device=q.device)
        i = torch.empty((0, k), dtype=torch.int64, device=q.device)
    pad_s = torch.full((most, k), float("-inf"), dtype=torch.float32, device=s.device)
    pad_i = torch.full((most, k), -1, dtype=torch.int64, device=s.device)
    pad_s[: s.shape[0]], pad_i[: i.shape[0]] = s, i
    from . import ops
    all_s = ops.all_gather(pad_s, group).view(G * most, k)
    all_i = ops.all_gather(pad_i, group).view(G * most, k)
    keep = torch.cat([torch.arange(g * most, g * most + bounds[g + 1] - bounds[g], device=s.device) for g in range(G)])
    ops.comm_check(s.device)                                       # a timed-out exchange is an error, not a result
    return all_s[keep].contiguous(), all_i[keep].contiguous()
