"""In-process multi-GPU PQ indexes: what `load_index_to_gpu(index, single_gpu_id=None)` returns.

Reference: models/repconc/evaluate_repconc.py:121-135 — `faiss.index_cpu_to_all_gpus(index, co)` with `co.shard = False`
puts a full copy of the index on EVERY visible GPU and Faiss splits each query batch across the copies;
run_repconc_eval.py:93-100 then searches from the main process only.  Here:

  * `ReplicatedPQIndex` — one `PQIndex` per device (codes and the permuted scan image copied device to device, over
    xGMI where peers are connected), each query batch split into contiguous slices, one slice per device, searched
    concurrently (one host thread per device: the search entry point ends with a status read) and concatenated in
    query order: exactly the results of a single-device search.
  * `ShardedPQIndex` — the rows are split instead (device d holds rows [off_d, off_d + n_d) with `id_offset = off_d`),
    every device scans its rows for all queries and the per-device top-k lists are merged with
    `sharded_search.merge_topk` ((score desc, id asc), the single-index order): what SURVEY §8e asks for next to the
    replica mode, and what makes an index larger than one GPU searchable.

Both are duck-typed like `PQIndex` (`search`, `ntotal`, `pq`, `codes`, `metric_type`, `set_centroids`, `add_codes`).
A device may be listed more than once (virtual replicas / shards on a single-GPU box, used by the tests).
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace
from typing import List, Optional, Sequence

import numpy as np
import torch

from .index import PQIndex
from .sharded_search import merge_topk


def _visible_devices() -> List[int]:
    return list(range(torch.cuda.device_count()))


def _clone_to(index: PQIndex, device: int, rows: Optional[slice] = None, id_offset: int = 0) -> PQIndex:
    out = PQIndex(index.pq.d, index.pq.M, index.pq.nbits, index.metric_type, device=torch.device("cuda", device))
    out.set_centroids(index.pq.centroids)
    codes = index.codes if rows is None else index.codes[rows]
    if codes.shape[0]:
        out.add_codes(codes)                       # device-to-device copy; the scan image is rebuilt on the target
    out.id_offset = id_offset
    return out


class _FanoutCentroids:
    """`index.pq.centroids` of a multi-device index.  Every part keeps its own resident table, so an in-place write —
    the reference's `faiss.copy_array_to_vector(c, index.pq.centroids)` idiom (finetune_jpq.py:211-213) — has to reach
    ALL of them: `copy_` fans out through `set_centroids`; reads come from part 0 (the parts are kept identical)."""

    def __init__(self, owner):
        self._owner = owner

    @property
    def _first(self) -> torch.Tensor:
        return self._owner.parts[0].pq.centroids

    shape = property(lambda self: self._first.shape)
    dtype = property(lambda self: self._first.dtype)
    device = property(lambda self: self._first.device)

    def numel(self):
        return self._first.numel()

    def copy_(self, src, non_blocking: bool = False):
        self._owner.set_centroids(src.reshape(self._first.shape))
        return self

    def detach(self):
        return self._first.detach()

    def reshape(self, *shape):
        return self._first.reshape(*shape)

    def cpu(self):
        return self._first.cpu()

    def __array__(self, dtype=None, copy=None):
        a = self._first.cpu().numpy()
        return a.astype(dtype) if dtype is not None else a


class _MultiIndex:
    def __init__(self, parts: List[PQIndex]):
        assert parts
        self.parts = parts
        p0 = parts[0].pq
        self.pq = SimpleNamespace(d=p0.d, M=p0.M, nbits=p0.nbits, code_size=p0.code_size, ksub=p0.ksub, dsub=p0.dsub,
                                  centroids=_FanoutCentroids(self))
        self.metric_type = parts[0].metric_type
        self.is_trained = parts[0].is_trained
        self.device = parts[0].device
        self._pool = ThreadPoolExecutor(max_workers=len(parts))

    @property
    def d(self):
        return self.pq.d

    def set_centroids(self, centroids):
        for p in self.parts:
            p.set_centroids(centroids)
        self.is_trained = True

    def _run(self, jobs):
        """jobs: list of (part, queries on any device, k) -> list of (scores, ids) on the part's device."""
        def one(job):
            part, q, k = job
            with torch.cuda.device(part.device):
                s, i = part.search(q.to(part.device, non_blocking=True), k)
                torch.cuda.current_stream(part.device).synchronize()
                return s, i
        if len(jobs) == 1:
            return [one(jobs[0])]
        return list(self._pool.map(one, jobs))

    @staticmethod
    def _as_tensor(x):
        as_numpy = not isinstance(x, torch.Tensor)
        q = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if as_numpy else x.float()
        return q, as_numpy


class ReplicatedPQIndex(_MultiIndex):
    """Full copy per device, queries split (the reference's co.shard = False)."""

    def __init__(self, index: PQIndex, devices: Optional[Sequence[int]] = None):
        devices = list(devices) if devices is not None else _visible_devices()
        parts = []
        for d in devices:
            same = index.device.index == d and not any(p is index for p in parts)
            parts.append(index if same else _clone_to(index, d, id_offset=index.id_offset))
        super().__init__(parts)

    @property
    def ntotal(self):
        return self.parts[0].ntotal

    @property
    def codes(self):
        return self.parts[0].codes

    def add_codes(self, new_codes):
        for p in self.parts:
            p.add_codes(new_codes if not isinstance(new_codes, torch.Tensor) else new_codes.to(p.device))

    def search(self, x, k: int):
        q, as_numpy = self._as_tensor(x)
        k = int(k)
        nq = q.shape[0]
        G = len(self.parts)
        bounds = [(nq * i) // G for i in range(G + 1)]
        jobs = [(self.parts[i], q[bounds[i]:bounds[i + 1]], k) for i in range(G) if bounds[i + 1] > bounds[i]]
        if not jobs:
            jobs = [(self.parts[0], q, k)]
        res = self._run(jobs)
        home = self.parts[0].device
        scores = torch.cat([r[0].to(home) for r in res], 0)
        ids = torch.cat([r[1].to(home) for r in res], 0)
        if as_numpy:
            return scores.cpu().numpy(), ids.cpu().numpy()
        return scores.to(q.device) if q.is_cuda else scores, ids.to(q.device) if q.is_cuda else ids


class ShardedPQIndex(_MultiIndex):
    """Rows split across devices, every device searches all queries, top-k lists merged."""

    def __init__(self, index: PQIndex, devices: Optional[Sequence[int]] = None):
        devices = list(devices) if devices is not None else _visible_devices()
        G, N = len(devices), index.ntotal
        bounds = [(N * i) // G for i in range(G + 1)]
        parts = [_clone_to(index, d, slice(bounds[i], bounds[i + 1]), index.id_offset + bounds[i])
                 for i, d in enumerate(devices)]
        super().__init__(parts)

    @property
    def ntotal(self):
        return sum(p.ntotal for p in self.parts)

    @property
    def codes(self):
        home = self.parts[0].device
        return torch.cat([p.codes.to(home) for p in self.parts], 0)

    def add_codes(self, new_codes):
        """Appended rows go to the last shard (ids stay contiguous)."""
        last = self.parts[-1]
        last.add_codes(new_codes if not isinstance(new_codes, torch.Tensor) else new_codes.to(last.device))

    def search(self, x, k: int):
        q, as_numpy = self._as_tensor(x)
        res = self._run([(p, q, int(k)) for p in self.parts])
        home = self.parts[0].device
        scores, ids = merge_topk(torch.stack([r[0].to(home) for r in res]), torch.stack([r[1].to(home) for r in res]), int(k))
        if as_numpy:
            return scores.cpu().numpy(), ids.cpu().numpy()
        return scores.to(q.device) if q.is_cuda else scores, ids.to(q.device) if q.is_cuda else ids


class ReplicatedIVFPQIndex:
    """The IVF index of BASELINE configs[3] ("IVF nlist=5000 ADC search at 8 GPUs") in one process: a full copy of the
    `IVFPQIndex` per device, every query batch split into contiguous slices searched concurrently, results concatenated
    in query order — what `ReplicatedPQIndex` does for the flat index, with `search(x, k, nprobe)`."""

    def __init__(self, index, devices: Optional[Sequence[int]] = None):
        devices = list(devices) if devices is not None else _visible_devices()
        self.parts = []
        for d in devices:
            same = index.device.index == d and not any(p is index for p in self.parts)
            self.parts.append(index if same else index.to(torch.device("cuda", d)))
        self.device, self.d, self.M, self.nlist = index.device, index.d, index.M, index.nlist
        self.nprobe = getattr(index, "nprobe", 8)         # Faiss's attribute: cells probed when search() is not told
        self.whole_query_set = True                       # batch_search: every query in one call (each replica takes its slice)
        self._pool = ThreadPoolExecutor(max_workers=len(self.parts))

    @property
    def ntotal(self):
        return self.parts[0].ntotal

    def search(self, x, k: int, nprobe: Optional[int] = None, method: str = "auto"):
        if nprobe is None:
            nprobe = self.nprobe
        as_numpy = not isinstance(x, torch.Tensor)
        q = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if as_numpy else x.float()
        nq, G = q.shape[0], len(self.parts)
        bounds = [(nq * i) // G for i in range(G + 1)]
        jobs = [(self.parts[i], q[bounds[i]:bounds[i + 1]]) for i in range(G) if bounds[i + 1] > bounds[i]] or [(self.parts[0], q)]

        def one(job):
            part, qq = job
            with torch.cuda.device(part.device):
                s, i = part.search(qq.to(part.device, non_blocking=True), int(k), int(nprobe), method)
                torch.cuda.current_stream(part.device).synchronize()
                return s, i
        res = [one(jobs[0])] if len(jobs) == 1 else list(self._pool.map(one, jobs))
        home = self.parts[0].device
        scores = torch.cat([r[0].to(home) for r in res], 0)
        ids = torch.cat([r[1].to(home) for r in res], 0)
        if as_numpy:
            return scores.cpu().numpy(), ids.cpu().numpy()
        return (scores.to(q.device), ids.to(q.device)) if q.is_cuda else (scores, ids)
