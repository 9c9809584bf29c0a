"""Batch-sharded constrained assignment: the `dist.is_initialized()` branch of RepCONC.quantize.

Reference: models/repconc/modeling_repconc.py:78-80 (all_reduce MAX/MIN of the per-m distance
range) and :149-157 (all_reduce SUM of the total and, every iteration, of the row sums).  Each
rank holds an equal row block of the batch (the recipes use --dataloader_drop_last); the
uniform-assignment constraint is over the GLOBAL batch.

MI355X mapping (SURVEY.md §5 "Distributed comm backend"): per Sinkhorn iteration every rank
contributes its [M,256] fp64 row sums (98 KB at M=48); they are ALL-GATHERED and summed locally
in rank order, so every rank computes bit-identical potentials and the codes cannot diverge
between ranks.  The global total of :148-152 cancels in the argmax and needs no collective.
Collectives go through torch.distributed (backend "nccl" = RCCL over xGMI) on the same stream
as the kernels.

The driver below is written against two small interfaces so the same choreography runs
  * on GPUs with the HIP kernels (`HipStages`, the product path),
  * as G virtual shards inside one process (`VirtualComm`, used on single-GPU boxes), and
  * under gloo on CPU in the test-suite, where tests inject a numpy stand-in for the stages.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


# ----------------------------------------------------------------------------- communicators
class TorchDistComm:
    """One rank of a torch.distributed process group."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def allreduce_minmax_(self, minmax: torch.Tensor, M: int):
        dist.all_reduce(minmax[:M], op=dist.ReduceOp.MAX, group=self.group)    # :79
        dist.all_reduce(minmax[M:], op=dist.ReduceOp.MIN, group=self.group)    # :80

    def allgather(self, t: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.group)
        return out

    def allgather_async(self, t: torch.Tensor):
        """Start the all-gather and return a callable that (stream-)waits for it and yields the [G,...] result,
        so that kernels launched in between overlap the collective."""
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        work = dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.group, async_op=True)

        def wait():
            work.wait()
            return out
        return wait


class SingleComm:
    world, rank = 1, 0

    def allreduce_minmax_(self, minmax, M):
        pass

    def allgather(self, t):
        return t.unsqueeze(0)

    def allgather_async(self, t):
        return lambda: t.unsqueeze(0)


# ----------------------------------------------------------------------------- HIP stages
class HipStages:
    """The product implementation of the per-rank stages (librepconc_hip.so)."""

    def dist_table(self, x, centroids):
        from . import ops
        return ops.dist_table(x, centroids, with_minmax=True)

    def centre_(self, d, minmax):
        from . import ops
        return ops.centre_(d, minmax)

    def state(self, d):
        from . import ops
        return ops.SinkhornState(d)


# ----------------------------------------------------------------------------- driver
def _single_column_codes(x_local, M, dtype):
    """Global batch of ONE row: the first row normalisation (modeling_repconc.py:158) turns every
    entry into Q/Q = 1 exactly, the argmax is a K-way exact tie and torch returns index 0."""
    dev = x_local.device
    return (torch.zeros((x_local.shape[0], M), dtype=dtype, device=dev),
            torch.zeros((1,), dtype=torch.int32, device=dev))


def _native_default() -> bool:
    import os
    return os.environ.get("RC_DIST_NATIVE", "1") != "0"


def _split_default() -> bool:
    import os
    return os.environ.get("RC_SHARD_SPLIT", "1") != "0"


def assign_sinkhorn_sharded(x_local, centroids, eps: float, iters: int, comm, stages=None,
                            dtype=torch.int64, split=None):
    """Constrained codes for this rank's rows.  Returns (codes [B_local, M], flags).

    Default on GPUs: `ops.assign_sinkhorn_dist` (csrc/comm.hip) — dist table, range all-reduce, centring, all sweeps
    and their all-gathers enqueued from C on two streams.  The Python-staged loop below is the same choreography
    against the pluggable `stages` / `comm` interfaces (RC_DIST_NATIVE=0 selects it on GPUs; the CPU gloo tests
    drive it with a numpy stand-in).

    With more than one rank the M sub-quantisers are solved as two independent halves in lock-step
    (`split`): while the row sums of one half are being all-gathered, the sweep of the other half runs, so
    the collective's latency (98 KB per rank, latency-bound on xGMI) is off the critical path.  The halves never
    interact — sub-quantisers are independent problems — so the codes are exactly those of the unsplit solve."""
    M = centroids.shape[0]
    if stages is None and isinstance(comm, TorchDistComm) and _native_default():
        # product path: the whole solve (RCCL included) behind one C call — no Python between iterations
        from . import ops
        ops.comm_init(comm.group)
        return ops.assign_sinkhorn_dist(x_local, centroids, eps, iters, dtype)
    stages = stages or HipStages()
    if comm.world * x_local.shape[0] == 1:
        return _single_column_codes(x_local, M, dtype)
    if split is None:
        split = comm.world > 1 and _split_default()
    d, minmax = stages.dist_table(x_local, centroids)
    comm.allreduce_minmax_(minmax, M)
    stages.centre_(d, minmax)
    if not split or M < 2:
        st = stages.state(d)
        rows = st.sweep(eps, 0, None)
        for t in range(1, iters):
            rows = st.sweep(eps, t, comm.allgather(rows))
        return st.argmax(eps, iters, comm.allgather(rows), dtype), st.flags
    h = M // 2
    sts = [stages.state(d[:h]), stages.state(d[h:])]
    pend = [comm.allgather_async(st.sweep(eps, 0, None)) for st in sts]
    for t in range(1, iters):
        for i, st in enumerate(sts):
            pend[i] = comm.allgather_async(st.sweep(eps, t, pend[i]()))
    codes = torch.cat([st.argmax(eps, iters, pend[i](), dtype) for i, st in enumerate(sts)], dim=1)
    return codes, sts[0].flags | sts[1].flags


def assign_sinkhorn_virtual(x_shards: Sequence, centroids, eps: float, iters: int, stages=None,
                            dtype=torch.int64, split: bool = False) -> List:
    """Run G shards of one batch in lock-step inside ONE process: the same stage calls and the
    same rank-ordered reduction as `assign_sinkhorn_sharded`, with the collectives replaced by
    local max/min/stack.  Used where only one device is visible (RCCL refuses two ranks on one
    GPU) and by the `sharded == unsharded` parity tests.  `split` solves the two halves of M separately,
    as the multi-rank driver does."""
    stages = stages or HipStages()
    M = centroids.shape[0]
    if sum(x.shape[0] for x in x_shards) == 1:
        outs = [_single_column_codes(x, M, dtype) for x in x_shards]
        return [o[0] for o in outs], [o[1] for o in outs]
    tabs = [stages.dist_table(x, centroids) for x in x_shards]
    mm = tabs[0][1].clone()
    for _, other in tabs[1:]:
        mm[:M] = torch.maximum(mm[:M], other[:M])
        mm[M:] = torch.minimum(mm[M:], other[M:])
    for d, _ in tabs:
        stages.centre_(d, mm)
    parts = [(0, M)] if not split or M < 2 else [(0, M // 2), (M // 2, M)]
    codes_parts, flags = [], [None] * len(tabs)
    for a, b in parts:
        states = [stages.state(d[a:b]) for d, _ in tabs]
        gathered = None
        for t in range(iters):
            gathered = torch.stack([st.sweep(eps, t, gathered) for st in states], dim=0)
        codes_parts.append([st.argmax(eps, iters, gathered, dtype) for st in states])
        for r, st in enumerate(states):
            flags[r] = st.flags if flags[r] is None else (flags[r] | st.flags)
    codes = [torch.cat([cp[r] for cp in codes_parts], dim=1) for r in range(len(tabs))]
    return codes, flags
