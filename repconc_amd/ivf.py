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
import os
from typing import Optional

import numpy as np
import torch

from . import _lib, ops
from .index import PQIndex


def coarse_kmeans(x: torch.Tensor, nlist: int, iters: int = 10, seed: int = 1234, chunk: int = 1 << 20) -> torch.Tensor:
    """Lloyd k-means of `nlist` centroids on x [n, D] (device): assignment by the fp32-MFMA GEMM + argmin kernel
    (rc_ivf_coarse_assign), centroid update by rc_ivf_coarse_update (stable counting sort of the rows by cell, one block per
    cell sums its rows in ascending order in fp64; an empty cell takes a counter-based random row) — nothing is read back
    between the iterations and the result is the same run to run.  Initial centroids: `nlist` rows of a seeded permutation."""
    n, D = x.shape
    rng = np.random.default_rng(seed)
    cent = x[torch.from_numpy(rng.permutation(n)[:nlist].copy()).to(x.device)].clone().float().contiguous()
    xt = ops._rows_f32(x)
    native = x.is_cuda and D % 16 == 0 and D <= 1024 and nlist <= 16384
    if native:
        lib, h, s, _ = ops._ctx(xt)
        wsb = lib.rc_ivf_coarse_update_ws_bytes(n, nlist)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=xt.device)
    for it in range(iters):
        if native:
            assign = coarse_assign(xt, cent, chunk, as_int32=True)
            _lib.check(lib.rc_ivf_coarse_update(h, C.c_void_p(xt.data_ptr()), xt.stride(0), C.c_void_p(assign.data_ptr()), n, D,
                                                nlist, C.c_void_p(cent.data_ptr()), None, seed, it, C.c_void_p(ws.data_ptr()), wsb,
                                                s), "rc_ivf_coarse_update", h)
            continue
        # widths / list counts outside the kernels' range (toy fixtures): the same step with library calls
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


def coarse_assign(x: torch.Tensor, cent: torch.Tensor, chunk: int = 1 << 20, as_int32: bool = False) -> torch.Tensor:
    """L2-nearest coarse centroid of every row, argmin_l (||c_l||^2 - 2 <x, c_l>), first minimum: the fp32-MFMA
    GEMM + fused argmin of csrc/ivf_search.hip (rc_ivf_coarse_assign); the [n, nlist] scores are never materialised."""
    ops._need_cuda(x, cent)
    xt = ops._rows_f32(x)
    cent = cent.float().contiguous()
    n, D = xt.shape
    nlist = cent.shape[0]
    if D % 16 != 0:
        # the MFMA kernel stages K in chunks of 16 (RC_ESHAPE otherwise); widths the PQ path cannot have anyway (toy
        # fixtures): the same argmin through a library GEMM, chunked so the [chunk, nlist] scores stay small
        c2 = (cent * cent).sum(1)
        parts = [torch.argmin(c2[None, :] - 2.0 * (xt[i:i + (1 << 16)] @ cent.T), dim=1) for i in range(0, n, 1 << 16)]
        return torch.cat(parts) if parts else torch.empty((0,), dtype=torch.int64, device=xt.device)
    lib, h, s, _ = ops._ctx(xt)
    wsb = lib.rc_ivf_coarse_assign_ws_bytes(nlist)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=xt.device)
    out = torch.empty((n,), dtype=torch.int32, device=xt.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    for i in range(0, n, chunk):                        # chunks only bound the grid size, not the memory
        part = xt[i:i + chunk]
        _lib.check(lib.rc_ivf_coarse_assign(h, p(part), xt.stride(0), p(cent), part.shape[0], D, nlist,
                                            C.c_void_p(out.data_ptr() + 4 * i), p(ws), wsb, s), "rc_ivf_coarse_assign", h)
    return out if as_int32 else out.to(torch.int64)


class IVFPQIndex:
    def __init__(self, d: int, M: int, nlist: int, device: Optional[torch.device] = None):
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type == "cuda" and self.device.index is None:        # "cuda" -> the current device, by index
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.d, self.M, self.nlist = d, M, nlist
        self.coarse = None                                                  # [nlist, d]
        self.pq_centroids = torch.zeros((M, 256, d // M), dtype=torch.float32, device=self.device)
        self.codes = torch.empty((0, M), dtype=torch.uint8, device=self.device)
        self.ids = torch.empty((0,), dtype=torch.int64, device=self.device)
        self.image = None            # permuted copy of `codes` for the conflict-free screen (list-centric search)
        self.image16 = None          # ... for its 16-query form (built on first use: whole-query-set calls at nprobe >= ~16)
        self.list_off = torch.zeros((nlist + 1,), dtype=torch.int64, device=self.device)
        self.ntotal = 0
        self._sizes_desc = np.zeros(0, np.int64)      # cell sizes, descending (host copy; set_lists fills it)
        self.nprobe = 8                               # Faiss's attribute: cells probed when search() is not told
        # batch_search (models/repconc/evaluate_repconc.py) hands an index with this attribute the WHOLE query set in one
        # call: a probed cell is scanned once for all the queries that probe it (6 980 dev queries at nprobe 8: ~11 per probed
        # cell instead of ~2 in 1 200-query batches — the screen's 8 gather columns fill up)
        self.whole_query_set = True

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
        self._sizes_desc = np.sort(cnt.cpu().numpy())[::-1].astype(np.int64)     # host copy: bounds the sample array
        self.image = None
        self.image16 = None
        if ops.adc_image_supported(self.M) and self.ntotal:
            self.image = torch.empty((ops.adc_image_rows_bytes(self.ntotal, self.M),), dtype=torch.uint8, device=self.device)
            ops.adc_scan_image_(self.codes, self.image, layout="rows")        # blocked by chunks of 16 rows: cells start anywhere

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

    def to(self, device) -> "IVFPQIndex":
        """A full copy of the index on another device (device-to-device copies, over xGMI where peers are connected)."""
        device = torch.device(device)
        out = IVFPQIndex(self.d, self.M, self.nlist, device=device)
        out.coarse = None if self.coarse is None else self.coarse.to(device)
        out.pq_centroids = self.pq_centroids.to(device).clone()
        out.codes, out.ids = self.codes.to(device), self.ids.to(device)
        out.list_off = self.list_off.to(device)
        out.ntotal = self.ntotal
        out._sizes_desc = self._sizes_desc
        out.nprobe = self.nprobe
        if self.image is not None:                      # the permutation of row n depends on n mod 16 only: copy, not rebuild
            out.image = self.image.to(device)
        if self.image16 is not None:
            out.image16 = self.image16.to(device)
        if device == self.device:                       # same device (virtual replica in the tests): real copies
            out.codes, out.ids, out.list_off = out.codes.clone(), out.ids.clone(), out.list_off.clone()
            out.image = None if out.image is None else out.image.clone()
            out.image16 = None if out.image16 is None else out.image16.clone()
            out.coarse = None if out.coarse is None else out.coarse.clone()
        return out

    # ---- search
    def probe(self, q: torch.Tensor, nprobe: int, ordered: bool = True) -> torch.Tensor:
        """[nq, nprobe] int32: the nprobe cells with the largest <q, centroid> (ties at the boundary: lower cell id).
        ordered=True ranks them by decreasing score (ties: lower cell id); ordered=False — what the searches use — returns
        the same set in ascending cell order from one HIP kernel (rc_ivf_select_probes) after the library GEMM."""
        s = q.float() @ self.coarse.T
        if not ordered and self.nlist <= 16384 and q.shape[0] > 0:
            s = s.contiguous()
            out = torch.empty((q.shape[0], nprobe), dtype=torch.int32, device=s.device)
            lib, h, st, _ = ops._ctx(s)
            _lib.check(lib.rc_ivf_select_probes(h, C.c_void_p(s.data_ptr()), s.shape[0], self.nlist, int(nprobe),
                                                C.c_void_p(out.data_ptr()), st), "rc_ivf_select_probes", h)
            return out
        if nprobe * 4 <= self.nlist:
            # select first, then order the selected cells by (score desc, cell asc): a full stable sort of nlist scores
            # per query costs more than the search itself at small nprobe
            top = torch.topk(s, nprobe, dim=1, sorted=False).indices
            top = torch.sort(top, dim=1).values
            sel = torch.gather(s, 1, top)
            order = torch.gather(top, 1, torch.argsort(sel, dim=1, descending=True, stable=True))
        else:
            order = torch.argsort(s, dim=1, descending=True, stable=True)[:, :nprobe]
        return order.to(torch.int32).contiguous()

    def search(self, x, k: int, nprobe: Optional[int] = None, method: str = "auto"):
        """nprobe: cells probed per query (default: the index's `nprobe` attribute, as with a Faiss IVF index).
        method: "lists" = list-centric 8-bit screen (rc_ivf_search_probes_q / _q16: the queries probing a cell share its read,
        M in {16,32,48,64,96}; tasks of 8 or — when a probed cell is shared by >= WIDE_MIN_SHARE queries of the call — 16
        queries; "lists8" / "lists16" force the width); "scan" = the per-query exact scan (rc_ivf_search); "auto" = lists
        where available.
        Both return the same (scores, ids)."""
        as_numpy = not isinstance(x, torch.Tensor)
        q = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if as_numpy else x).to(self.device, torch.float32)
        q = q.contiguous()
        nq = q.shape[0]
        if nprobe is None:
            nprobe = self.nprobe
        if nq > self.MAX_QUERY_BATCH:                      # the C entries index their per-query workspaces with 32 bits
            parts = [self.search(q[i:i + self.MAX_QUERY_BATCH], k, nprobe, method) for i in range(0, nq, self.MAX_QUERY_BATCH)]
            scores, ids = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
            return (scores.cpu().numpy(), ids.cpu().numpy()) if as_numpy else (scores, ids)
        nprobe = min(int(nprobe), self.nlist)
        probes = self.probe(q, nprobe, ordered=False)
        if method not in ("auto", "lists", "lists8", "lists16", "lists_host_plan", "scan"):
            raise ValueError("method must be auto|lists|lists8|lists16|lists_host_plan|scan")
        if method in ("lists", "lists8", "lists16", "lists_host_plan") and self.image is None:
            raise _lib.RepconcHipError(f"the list-centric search needs M in (16, 32, 48, 64, 96), not {self.M}")
        if method == "auto" and self.image is not None:
            # few probed rows per query: the per-query scan has less fixed work (task list, per-query byte tables)
            method = "lists" if self.ntotal * nprobe / max(self.nlist, 1) * self.M >= self.LISTS_MIN_BYTES else "scan"
        if method in ("lists", "lists8", "lists16", "lists_host_plan") and nq > 0:
            if method == "lists_host_plan":
                scores, ids = self._search_lists_host_plan(q, probes, int(k), nprobe)
            else:      # "lists": the screen's width by the call's queries per probed cell; lists8 / lists16 force it
                scores, ids = self._search_lists(q, probes, int(k), nprobe, width={"lists8": 8, "lists16": 16}.get(method))
            return (scores.cpu().numpy(), ids.cpu().numpy()) if as_numpy else (scores, ids)
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
            # more than 16384 probed rows tie at the k-th score (duplicated passages): the path without any list
            scores, ids = self._search_exact_probed(q, probes, int(k))
        if as_numpy:
            return scores.cpu().numpy(), ids.cpu().numpy()
        return scores, ids

    def _search_exact_probed(self, q: torch.Tensor, probes: torch.Tensor, k: int):
        """The answer for ANY index content (the flat search's `rc_adc_search_exact`, query by query, over the rows of the
        probed cells): exact scores of every probed row, the k best in (score desc, corpus id asc) order by radix select.
        Slow — one gather of the probed rows per query — and only reached when the faster paths cannot decide (thousands of
        identical rows in the probed cells)."""
        nq = q.shape[0]
        scores = torch.full((nq, k), float("-inf"), dtype=torch.float32, device=self.device)
        ids = torch.full((nq, k), -1, dtype=torch.int64, device=self.device)
        off = self.list_off
        for j in range(nq):
            cells = probes[j].long()
            a, b = off[cells], off[cells + 1]
            rows = torch.cat([torch.arange(int(x), int(y), device=self.device) for x, y in zip(a.tolist(), b.tolist())]) \
                if len(cells) else torch.empty(0, dtype=torch.int64, device=self.device)
            if rows.numel() == 0:
                continue
            corpus = self.ids[rows]
            order = torch.argsort(corpus)                        # local row order = corpus id order: the search's tie rule
            rows, corpus = rows[order], corpus[order]
            sub = self.codes[rows].contiguous()
            s, i = ops.adc_search_exact(sub, self.pq_centroids, q[j:j + 1], k)
            valid = i[0] >= 0
            scores[j] = s[0]
            ids[j, valid] = corpus[i[0][valid]]
        return scores, ids

    # ---- list-centric search
    SAMPLE_STEP = 8                 # every 8th (up to every 64th) row of a probed cell is scored exactly ...
    # head-room of the sampled threshold (standard deviations of the sample rank; ops.ADC_SEL_SLACK).  Round 6: 6 -> 4.  A query's
    # sample is ~1/32 of its probed rows (mu = 31 at k = 1000): at 4 no query of 41 880 was answered again by the scan and a 6 980-query
    # search takes 3.04 / 4.72 / 8.15 ms at nprobe 8 / 32 / 128 instead of 3.23 / 4.82 / 8.32; at 3 a handful are (4-8 of 41 880) and the
    # extra launches cost more than the smaller lists save (profiles/r06m_ivf_slack.txt)
    SEL_SLACK = 4.0
    # ... so that about this many sampled rows per query place the candidate threshold.  Round 6: 6144 -> 1536 — the exact fp32 scoring
    # of the sample (random 4-byte LDS gathers, two thirds of their cycles bank conflicts) was 1.24 of the 8.9 ms of a 6 980-query search
    # at nprobe 128; a quarter of the sample costs a few hundred more candidates per query and wins: ms per 6 980-query search at
    # nprobe 32 / 128: 4.71 -> 4.13, 8.32 -> 7.87; 1 200 queries: 1.08 -> 0.99, 1.59 -> 1.52; no query answered again by the scan
    # (profiles/r06m_ivf_sample_rows.txt)
    SAMPLE_ROWS = 1536
    CAND_CAP = 16384                # candidate keys per query (ADC_CAND_CAP)
    KEEP_ALL_ROWS = 4096            # queries probing no more rows than this re-score every row (no threshold)
    MAX_QUERY_BATCH = 16384         # queries per C call (rc_ivf_search_lists / _probes refuse more than 32768)
    LISTS_MIN_BYTES = 576000        # "auto": average probed code bytes per query from which the list-centric search pays (round 3, 1200 queries: the two meet at ~6.5 k rows for M = 96, ~10.6 k for M = 48)

    def _sample_step(self, nprobe: int) -> int:
        """about SAMPLE_ROWS exactly scored rows per query place the candidate threshold"""
        ss = self.SAMPLE_STEP
        while ss < 64 and self.ntotal * nprobe / max(self.nlist, 1) / (2 * ss) >= self.SAMPLE_ROWS:
            ss *= 2
        return ss

    # queries of a call per probed cell (nq x nprobe / nlist) from which the 16-query screen is used: below it most tasks would
    # hold <= 8 queries and pay the 16-query form's two table phases per 32 sub-quantisers for nothing.  [MI355X] M = 96, 8.84 M
    # rows in 5000 cells, ms per search 8-query -> 16-query (profiles/r06f_ivf_width_bench.txt): 6 980 queries, nprobe 8 / 32 /
    # 128 (11 / 45 / 179 queries per cell): 3.32 -> 3.15, 5.06 -> 4.94, 8.95 -> 8.54; 1 200 queries (1.9 / 7.7 / 31): 0.83 ->
    # 0.88, 1.11 -> 1.10, 1.81 -> 1.66
    WIDE_MIN_SHARE = 9.0

    def _wide_image(self):
        if self.image16 is None:
            self.image16 = torch.empty((ops.adc_image_rows_bytes(self.ntotal, self.M),), dtype=torch.uint8, device=self.device)
            ops.adc_scan_image_(self.codes, self.image16, layout="rows16")
        return self.image16

    def _search_lists(self, q: torch.Tensor, probes: torch.Tensor, k: int, nprobe: int, sel_slack: Optional[float] = None,
                      max_retries: int = 3, width: Optional[int] = None):
        """rc_ivf_search_probes_q / _q16: sample layout, ranks and the (cell, <= 8 | 16 queries) task list are made on the
        device.  width: 8, 16 or None = by the mean number of the call's queries per probed cell (WIDE_MIN_SHARE)."""
        dev, nq = self.device, q.shape[0]
        if width is None:
            env = os.environ.get("RC_IVF_WIDTH")
            width = int(env) if env in ("8", "16") else (16 if nq * nprobe / max(self.nlist, 1) >= self.WIDE_MIN_SHARE else 8)
        if width not in (8, 16):
            raise ValueError("width must be 8 or 16")
        image = self._wide_image() if width == 16 else self.image
        ss = self._sample_step(nprobe)
        top = self._sizes_desc[:nprobe]                                    # the nprobe largest cells bound a query's sample
        sstride = max(4, int((16 * (top // (16 * ss)) + np.minimum(top % (16 * ss), 16)).sum()))
        lut = ops.adc_lut(self.pq_centroids, q)
        lib, h = _lib.load(), _lib.handle(dev.index)
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        wsb = lib.rc_ivf_search_probes_ws_bytes(self.M, nq, nprobe, self.nlist, sstride)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
        ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        flags = torch.zeros((1 + nq,), dtype=torch.int32, device=dev)     # [status | per-query status]: one fill
        status, qstatus = flags[:1], flags[1:]
        p = lambda t: C.c_void_p(t.data_ptr())
        slack = float(self.SEL_SLACK if sel_slack is None else sel_slack)
        ops._warm_retry_ops(dev)          # the framework operators of the per-query repeat path, once per device
        if not getattr(self, "_scan_path_warm", False):
            # ... and the path that answers a flagged query (the per-query scan), once per index on one query: its first use in a
            # process cost ~100 ms of operator loading in the middle of a search (profiles/r06m_ivf_sample_rows.txt, first line of
            # nprobe 128), every later one well under a millisecond
            self._scan_path_warm = True
            self.search(q[:1], min(int(k), 16), nprobe, method="scan")
        for attempt in range(max_retries + 1):
            if attempt:
                flags.zero_()
            fn = lib.rc_ivf_search_probes_q16 if width == 16 else lib.rc_ivf_search_probes_q
            _lib.check(fn(h, p(self.codes), p(image), p(self.list_off), p(self.ids), self.ntotal, self.nlist, self.M, 256, p(lut),
                          nq, p(probes), nprobe, sstride, ss, int(k), slack, self.KEEP_ALL_ROWS, p(scores), p(ids), p(status),
                          p(qstatus), p(ws), wsb, s), "rc_ivf_search_probes_q16" if width == 16 else "rc_ivf_search_probes_q", h)
            st = int(status.item())
            if st == 0:
                return scores, ids
            bad = torch.nonzero(qstatus != 0).flatten()
            if not (st & 4) and bad.numel() > 0:
                # per-query status: the other queries' results stand; the flagged ones (degenerate cells: thousands of equal
                # scores around the threshold) are answered by the per-query exact scan — one bad query does not make the
                # batch repeat
                bs, bi = self.search(q[bad], k, nprobe, method="scan")
                scores[bad], ids[bad] = bs, bi
                return scores, ids
            # status bit 2: a survivor stream of the screen filled up (it may have dropped anybody's rows) -> fewer survivors, again
            slack = max(slack / 3.0, 0.0)
        # no slack fits the streams: the per-query exact scan decides
        return self.search(q, k, nprobe, method="scan")

    def _search_lists_host_plan(self, q: torch.Tensor, probes: torch.Tensor, k: int, nprobe: int, sel_slack: float = 6.0,
                                max_retries: int = 3):
        """The same search through rc_ivf_search_lists, with the plan spelled out in torch (what round 2 first shipped; kept
        as the readable statement of the plan and as the cross-check of the device-side planner in the tests)."""
        dev, nq = self.device, q.shape[0]
        pl = probes.long()
        sizes = (self.list_off[1:] - self.list_off[:-1])[pl]                                   # [nq, nprobe]
        rows = sizes.sum(1)
        # sample step: about SAMPLE_ROWS exactly scored rows per query place the candidate threshold
        ss = self._sample_step(nprobe)
        ssz = 16 * (sizes // (16 * ss)) + torch.clamp(sizes % (16 * ss), max=16)               # sampled rows per probe (runs of 16)
        scs = torch.cumsum(ssz, 1)
        sbase = (scs - ssz).to(torch.int32).contiguous()
        scount = scs[:, -1]
        # tasks: (cell, <= 8 of the queries probing it); bookkeeping on [nq * nprobe] pairs
        flat_cell = pl.reshape(-1)
        order = torch.argsort(flat_cell, stable=True)
        sorted_q = (order // nprobe).to(torch.int32).contiguous()
        per_cell = torch.bincount(flat_cell, minlength=self.nlist)
        cell_start = torch.cumsum(per_cell, 0) - per_cell
        tasks_per_cell = (per_cell + 7) // 8
        ntasks = int(tasks_per_cell.sum().item())
        task_cell = torch.repeat_interleave(torch.arange(self.nlist, device=dev), tasks_per_cell)
        first_task = torch.cumsum(tasks_per_cell, 0) - tasks_per_cell
        within = torch.arange(ntasks, device=dev) - first_task[task_cell]
        task_qstart = (cell_start[task_cell] + 8 * within).to(torch.int32).contiguous()
        task_qcnt = torch.clamp(per_cell[task_cell] - 8 * within, max=8).to(torch.int32).contiguous()
        task_list = task_cell.to(torch.int32).contiguous()
        sstride = max(4, int(scount.max().item()))
        lut = ops.adc_lut(self.pq_centroids, q)
        lib, h = _lib.load(), _lib.handle(dev.index)
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        wsb = lib.rc_ivf_search_lists_ws_bytes(self.M, nq, sstride)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
        ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        status = torch.zeros((1,), dtype=torch.int32, device=dev)
        rows32, scount32 = rows.to(torch.int32).contiguous(), scount.to(torch.int32).contiguous()
        p = lambda t: C.c_void_p(t.data_ptr())
        slack = float(sel_slack)
        for _ in range(max_retries + 1):
            # rank of the sample score used as threshold: mu = expected number of the k best among the sampled rows;
            # queries whose probed rows fit the candidate list keep every row (rank 0 -> threshold -inf)
            mu = k * scount.double() / rows.clamp_min(1).double()
            rank = (mu + slack * torch.sqrt(mu + 1.0) + 4.0).floor() + 1
            cap = (0.8 * self.CAND_CAP * scount.double() / rows.clamp_min(1).double()).floor()
            rank = torch.where((rank > cap) & (cap >= mu + 2.5 * torch.sqrt(mu + 1.0) + 2.0), cap, rank)
            rank = torch.where(rows <= self.KEEP_ALL_ROWS, torch.zeros_like(rank), torch.minimum(rank, scount.double()))
            rank32 = rank.to(torch.int32).contiguous()
            status.zero_()
            _lib.check(lib.rc_ivf_search_lists(h, p(self.codes), p(self.image), p(self.list_off), p(self.ids), self.ntotal,
                                               self.M, 256, p(lut), nq, p(probes), p(sbase), p(scount32), p(rows32), p(rank32),
                                               nprobe, sstride, ss, p(task_list), p(task_qstart), p(task_qcnt), p(sorted_q),
                                               ntasks, k, p(scores), p(ids), p(status), p(ws), wsb, s),
                       "rc_ivf_search_lists", h)
            st = int(status.item())
            if st == 0:
                return scores, ids
            slack = max(slack, 0.0) * 3.0 + 2.0 if (st & 1) else max(slack / 3.0, 0.0)
        return self.search(q, k, nprobe, method="scan")
