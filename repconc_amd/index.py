"""PQ index resident in HBM + ADC search: the Faiss objects of the reference's evaluation path.

The reference builds `faiss.IndexPQ(D, M, 8, METRIC_INNER_PRODUCT)`, appends raw uint8 codes,
wraps it as a 1-list IndexIVFPQ, clones it to the GPU(s) and calls `.search`
(models/repconc/evaluate_repconc.py:78-135,180-185; models/jpq/finetune_jpq.py:157-161,176,209-214).
`PQIndex` is duck-typed like those objects (`search`, `ntotal`, `pq.{d,M,nbits,code_size,ksub,dsub,
centroids}`, `codes`, `metric_type`) so the call sites keep working, but:

  * codes live in ONE uint8 [N, M] device tensor that is appended to in place (amortised
    doubling) instead of being round-tripped through numpy on every add (evaluate_repconc.py:94-97);
  * `set_centroids` rewrites only the M*256*dsub table (786 KB) — JPQ's per-step "re-clone the
    whole index to the GPU" (finetune_jpq.py:209-214) becomes a 786 KB device copy;
  * there is no coarse quantiser: the reference's IVFPQ has nlist=1 with a zero centroid, i.e. it
    IS a flat PQ scan (SURVEY.md §6 discrepancy note).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import ops

METRIC_INNER_PRODUCT = 0  # faiss.METRIC_INNER_PRODUCT


class PQIndex:
    def __init__(self, d: int, M: int, nbits: int = 8, metric=METRIC_INNER_PRODUCT,
                 device: Optional[torch.device] = None):
        assert nbits == 8, "256 centroids per sub-quantiser (evaluate_repconc.py:80)"
        assert d % M == 0
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type == "cuda" and self.device.index is None:        # "cuda" -> the current device, by index
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.metric_type = metric
        self.is_trained = False
        self.ntotal = 0
        self.id_offset = 0       # global id of local row 0 (row-sharded indexes)
        self.sel_slack = ops.ADC_SEL_SLACK    # head-room of the sampled candidate threshold, in standard deviations of the rank
        self.last_search = None
        self._centroids = torch.zeros((M, 256, d // M), dtype=torch.float32, device=self.device)
        self._codes = torch.empty((0, M), dtype=torch.uint8, device=self.device)
        # permuted copy of the codes streamed by the conflict-free ADC screen (csrc/adc_search.hip), kept in step
        # with `_codes` row for row; None for the M that do not use one
        self._image = (torch.empty((0,), dtype=torch.uint8, device=self.device)        # flat buffer, ops.adc_image_bytes
                       if ops.adc_image_supported(M) else None)
        self.pq = SimpleNamespace(d=d, M=M, nbits=nbits, code_size=M, ksub=256, dsub=d // M, centroids=self._centroids)

    # ---- Faiss-like attributes
    @property
    def codes(self) -> torch.Tensor:
        """uint8 [ntotal, M] view of the resident codes."""
        return self._codes[: self.ntotal]

    @property
    def d(self):
        return self.pq.d

    def set_centroids(self, centroids):
        """initialize_index's `copy_array_to_vector(centroids.ravel(), index.pq.centroids)`,
        evaluate_repconc.py:84-85 ([m][k][j] order)."""
        c = centroids.detach() if isinstance(centroids, torch.Tensor) else torch.from_numpy(np.asarray(centroids))
        self._centroids.copy_(c.reshape(self._centroids.shape).to(self.device, torch.float32))
        self.is_trained = True

    def add_codes(self, new_codes):
        """add_docs, evaluate_repconc.py:89-98."""
        c = new_codes if isinstance(new_codes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(new_codes))
        assert c.dim() == 2 and c.shape[1] == self.pq.M, (tuple(c.shape), self.pq.M)
        if c.dtype != torch.uint8:
            c = c.to(torch.uint8)
        n = c.shape[0]
        need = self.ntotal + n
        if need > self._codes.shape[0]:
            grown = torch.empty((max(need, int(self._codes.shape[0] * 1.5)), self.pq.M), dtype=torch.uint8,
                                device=self.device)
            grown[: self.ntotal] = self._codes[: self.ntotal]
            self._codes = grown
            if self._image is not None:
                # the image's layout does not depend on the capacity: the bytes of the rows held so far move as they are
                gi = torch.empty((ops.adc_image_bytes(grown.shape[0], self.pq.M),), dtype=torch.uint8, device=self.device)
                held = ops.adc_image_bytes(self.ntotal, self.pq.M)
                gi[:held] = self._image[:held]
                self._image = gi
        self._codes[self.ntotal:need] = c.to(self.device)
        if self._image is not None and n > 0:
            ops.adc_scan_image_(self._codes, self._image, self.ntotal, n)
        self.ntotal = need

    def add(self, x):
        """IndexPQ.add: encode by L2-nearest sub-centroid and append."""
        xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        self.add_codes(ops.assign_nearest(xt.to(self.device), self._centroids, torch.uint8))

    # ---- search
    def search(self, x, k: int):
        """(scores [nq,k], ids [nq,k]); numpy in -> numpy out (evaluate_repconc.py:182), CUDA tensors
        in -> CUDA tensors out (finetune_jpq.py:176 via faiss.contrib.torch_utils)."""
        return self.search_async(x, k)()

    def search_async(self, x, k: int, stats=None):
        """Enqueue the search and return a callable that yields what `search` returns.  Nothing synchronises with the
        host until it is called, so a caller with several query batches (`batch_search`) can enqueue them all first: the
        device then never idles between batches waiting for the host to read a status word and launch the next one."""
        as_numpy = not isinstance(x, torch.Tensor)
        q = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if as_numpy else x
        q = q.to(self.device, torch.float32, non_blocking=True)
        # prefixes of the row-major buffers are contiguous views: nothing is copied
        pending = ops.adc_search(self._codes[: self.ntotal], self._centroids, q, int(k), id_offset=self.id_offset,
                                 scan_image=self._image, defer=True, stats=stats, sel_slack=self.sel_slack)
        self.last_search = pending           # its .stats say how many queries had to be repeated / answered by the exact path

        def finish():
            scores, ids = pending.result()
            if as_numpy:
                return scores.cpu().numpy(), ids.cpu().numpy()
            return scores, ids
        return finish

    def reconstruct_n(self, i0: int, n: int) -> torch.Tensor:
        return ops.decode_raw(self.codes[i0:i0 + n].contiguous(), self._centroids)
