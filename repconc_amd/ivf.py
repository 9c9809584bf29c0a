"""IVF-PQ index (nlist > 1) — a build-side EXTENSION of the reference, which only ever wraps its PQ index as an
IVFPQ with ONE list and a zero coarse centroid (models/repconc/evaluate_repconc.py:101-118).  BASELINE.json's
"IVF nlist=5000" config and SURVEY.md §8d input D ask for it; there is no reference behaviour to match, so the
semantics are fixed here:

  * coarse quantiser: `nlist` centroids fitted by Lloyd k-means (L2) on (rotated) document embeddings — a library GEMM
    per assignment step (torch.mm on rocBLAS), bookkeeping in torch;
  * a document goes to its L2-nearest coarse centroid; its PQ code is the ORDINARY RepCONC code of the whole vector
    (`by_residual = False`), so codes produced by the model are stored unchanged;
  * a query probes the `nprobe` cells with the largest inner product <q, centroid> and scans only their rows with the
    exact ADC arithmetic of the flat index (`rc_ivf_search`); probing every cell returns exactly the flat result.

Rows are stored list-major: `codes` [N,M] sorted by cell, `list_off` [nlist+1], `ids` [N] original positions.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib, ops
from .index import PQIndex


def coarse_kmeans(x: torch.Tensor, nlist: int, iters: int = 10, seed: int = 1234, chunk: int = 1 << 16) -> torch.Tensor:
    """Lloyd k-means of `nlist` centroids on x [n, D] (device).  Empty cells are re-seeded from random points."""
    n, D = x.shape
    rng = np.random.default_rng(seed)
    cent = x[torch.from_numpy(rng.permutation(n)[:nlist].copy()).to(x.device)].clone().float()
    for _ in range(iters):
        assign = coarse_assign(x, cent, chunk)
        sums = torch.zeros_like(cent)
        sums.index_add_(0, assign, x.float())
        cnt = torch.bincount(assign, minlength=nlist).to(cent.dtype)
        nz = cnt > 0
        cent[nz] = sums[nz] / cnt[nz, None]
        empty = (~nz).nonzero().flatten()
        if len(empty):
            cent[empty] = x[torch.from_numpy(rng.integers(0, n, len(empty))).to(x.device)].float()
    return cent


def coarse_assign(x: torch.Tensor, cent: torch.Tensor, chunk: int = 1 << 20) -> torch.Tensor:
    """L2-nearest coarse centroid of every row, argmin_l (||c_l||^2 - 2 <x, c_l>), first minimum: the fp32-MFMA
    GEMM + fused argmin of csrc/ivf_search.hip (rc_ivf_coarse_assign); the [n, nlist] scores are never materialised."""
    xt = ops._rows_f32(x)
    cent = cent.float().contiguous()
    n, D = xt.shape
    nlist = cent.shape[0]
    lib, h, s, _ = ops._ctx(xt)
    wsb = lib.rc_ivf_coarse_assign_ws_bytes(nlist)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=xt.device)
    out = torch.empty((n,), dtype=torch.int32, device=xt.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    for i in range(0, n, chunk):                        # chunks only bound the grid size, not the memory
        part = xt[i:i + chunk]
        _lib.check(lib.rc_ivf_coarse_assign(h, p(part), xt.stride(0), p(cent), part.shape[0], D, nlist,
                                            C.c_void_p(out.data_ptr() + 4 * i), p(ws), wsb, s), "rc_ivf_coarse_assign", h)
    return out.to(torch.int64)


class IVFPQIndex:
    def __init__(self, d: int, M: int, nlist: int, device: Optional[torch.device] = None):
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.d, self.M, self.nlist = d, M, nlist
        self.coarse = None                                                  # [nlist, d]
        self.pq_centroids = torch.zeros((M, 256, d // M), dtype=torch.float32, device=self.device)
        self.codes = torch.empty((0, M), dtype=torch.uint8, device=self.device)
        self.ids = torch.empty((0,), dtype=torch.int64, device=self.device)
        self.list_off = torch.zeros((nlist + 1,), dtype=torch.int64, device=self.device)
        self.ntotal = 0

    # ---- build
    def train(self, x, iters: int = 10, seed: int = 1234):
        xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        self.coarse = coarse_kmeans(xt.to(self.device), self.nlist, iters, seed)

    def set_centroids(self, centroids):
        c = centroids.detach() if isinstance(centroids, torch.Tensor) else torch.from_numpy(np.asarray(centroids))
        self.pq_centroids.copy_(c.reshape(self.pq_centroids.shape).to(self.device, torch.float32))

    def set_lists(self, codes: torch.Tensor, list_ids: torch.Tensor):
        """Store `codes` [N,M] (corpus order) given each row's cell: stable sort by cell."""
        codes = codes.to(self.device)
        list_ids = list_ids.to(self.device)
        order = torch.argsort(list_ids, stable=True)
        self.codes = codes[order].contiguous()
        self.ids = order.to(torch.int64).contiguous()
        cnt = torch.bincount(list_ids, minlength=self.nlist)
        self.list_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=self.device), torch.cumsum(cnt, 0)]).contiguous()
        self.ntotal = codes.shape[0]

    def add(self, x, codes: Optional[torch.Tensor] = None):
        """Index (rotated) embeddings x [N,d]: nearest PQ codes (unless the model's `codes` are given) + coarse cell."""
        assert self.coarse is not None, "train() the coarse quantiser first"
        xt = (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))).to(self.device)
        if codes is None:
            codes = ops.assign_nearest(xt, self.pq_centroids, torch.uint8)
        self.set_lists(codes, coarse_assign(xt, self.coarse))

    @classmethod
    def from_flat(cls, flat: PQIndex, nlist: int, x=None, iters: int = 10, train_rows: int = 1 << 18):
        """IVF view of an existing flat PQ index.  Without embeddings the cells are fitted on (and rows assigned by)
        the reconstructions decode(codes) — what the reference's pipeline has at hand once the corpus is coded."""
        ivf = cls(flat.pq.d, flat.pq.M, nlist, device=flat.device)
        ivf.set_centroids(flat.pq.centroids)
        if x is None:
            chunks = [flat.reconstruct_n(i, min(1 << 18, flat.ntotal - i)) for i in range(0, flat.ntotal, 1 << 18)]
            xt = torch.cat(chunks, 0)
        else:
            xt = (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))).to(flat.device)
        sel = torch.from_numpy(np.sort(np.random.default_rng(1234).permutation(xt.shape[0])[:train_rows])).to(flat.device)
        ivf.coarse = coarse_kmeans(xt[sel], nlist, iters)
        ivf.set_lists(flat.codes, coarse_assign(xt, ivf.coarse))
        return ivf

    # ---- search
    def probe(self, q: torch.Tensor, nprobe: int) -> torch.Tensor:
        """[nq, nprobe] cells by decreasing <q, centroid> (ties: lower cell id), int32."""
        s = q.float() @ self.coarse.T
        order = torch.argsort(s, dim=1, descending=True, stable=True)[:, :nprobe]
        return order.to(torch.int32).contiguous()

    def search(self, x, k: int, nprobe: int):
        as_numpy = not isinstance(x, torch.Tensor)
        q = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if as_numpy else x).to(self.device, torch.float32)
        q = q.contiguous()
        nq = q.shape[0]
        nprobe = min(int(nprobe), self.nlist)
        probes = self.probe(q, nprobe)
        sizes = (self.list_off[1:] - self.list_off[:-1])[probes.long()]                      # [nq, nprobe]
        csum = torch.cumsum(sizes, 1)
        base = (csum - sizes).to(torch.int32).contiguous()
        count = csum[:, -1].to(torch.int32).contiguous()
        stride = max(4, int(count.max().item()))
        stride = (stride + 3) // 4 * 4
        lut = ops.adc_lut(self.pq_centroids, q)
        lib, h = _lib.load(), _lib.handle(self.device.index)
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        wsb = lib.rc_ivf_search_ws_bytes(nq, stride)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=self.device)
        scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        status = torch.zeros((1,), dtype=torch.int32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.rc_ivf_search(h, p(self.codes), p(self.list_off), p(self.ids), self.ntotal, self.M, 256, p(lut),
                                     p(probes), p(base), p(count), nq, nprobe, stride, int(k), p(scores), p(ids),
                                     p(status), p(ws), wsb, s), "rc_ivf_search", h)
        if int(status.item()) & 2:
            raise _lib.RepconcHipError("IVF search: more than 16384 rows tie at the k-th score")
        if as_numpy:
            return scores.cpu().numpy(), ids.cpu().numpy()
        return scores, ids
