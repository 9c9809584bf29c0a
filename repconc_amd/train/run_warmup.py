"""OPQ / PQ warm-up: fit the rotation and the M x 256 sub-centroids on corpus embeddings, then code the corpus.

Mirror of train/run_warmup.py:85-132 (`warmup_from_embeds(corpus_embeds, repconc) -> (repconc, index)`).  The
reference delegates everything to Faiss (`index_factory("OPQ{M},PQ{M}x8", IP)`, GPU flat-L2 assignment inside the
k-means, CPU centroid update).  Faiss is not available to pin against (parity unpinned); the
training loop below restates Faiss 1.7.x's published procedure:

  * OPQ: <= 65 536 training rows (seeded permutation, seed 1234); 50 outer iterations of
    {rotate, PQ k-means (40 Lloyd iterations the first time, 4 warm-started afterwards), encode, decode,
     orthogonal Procrustes: R = U V^T with X^T X_rec = U S V^T};
  * final PQ: 25 Lloyd iterations on the rotated training rows, random-sample initialisation, empty clusters
    re-seeded by splitting the largest cluster with a +-1/1024 perturbation;
  * codes of the whole corpus by L2-nearest sub-centroid.

Arithmetic: assignment (`rc_pq_assign_nearest`), sufficient statistics (`rc_kmeans_stats`) and centroid update
(`rc_kmeans_update`) are the HIP kernels; the 768x768 rotation GEMM and SVD are library calls on PyTorch-ROCm
(SURVEY.md §2.3 K8).  With several ranks each rank passes its corpus shard and the statistics are combined with one
all-gather + rank-ordered sum per Lloyd iteration (`gather_stats_`).
"""
from __future__ import annotations

import logging
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..index import METRIC_INNER_PRODUCT, PQIndex

logger = logging.getLogger(__name__)

MAX_TRAIN_POINTS = 256 * 256      # Faiss: max_points_per_centroid (256) x ksub
SEED = 1234


class PreTransformIndex:
    """The pair Faiss calls IndexPreTransform(OPQMatrix, IndexPQ): `.index` is the PQ index over ROTATED
    vectors (what run_warmup.py:187 writes out), `.A` the rotation [d_out, d_in] applied as x @ A.T."""

    def __init__(self, A: torch.Tensor, index: PQIndex):
        self.A = A
        self.index = index
        self.ntotal = index.ntotal

    def search(self, x, k):
        as_numpy = not isinstance(x, torch.Tensor)
        q = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) if as_numpy else x
        qr = (q.to(self.A.device, torch.float32) @ self.A.T).contiguous()
        scores, ids = self.index.search(qr, k)
        return (scores.cpu().numpy(), ids.cpu().numpy()) if as_numpy else (scores, ids)


def gather_stats_(sums: torch.Tensor, counts: torch.Tensor, group=None):
    """Sum the per-shard Lloyd statistics over the ranks, in place (BASELINE north_star: "RCCL all-gather
    over xGMI of per-shard centroid sufficient statistics").  ONE all-gather of the packed [M,K,dsub+1] fp64 block
    (sums | counts; 1.7 MB per rank at M=48) followed by a rank-ordered sum: unlike an all-reduce, whose ring order is
    not fixed, every rank then holds bit-identical statistics and therefore bit-identical centroids."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sums, counts
    G = dist.get_world_size(group)
    M, K, dsub = sums.shape
    packed = torch.cat([sums, counts.to(sums.dtype).unsqueeze(-1)], dim=-1).contiguous()     # counts < 2^53: exact
    flat = torch.empty((G * M, K, dsub + 1), dtype=packed.dtype, device=packed.device)    # concatenation form (gloo + nccl)
    dist.all_gather_into_tensor(flat, packed, group=group)
    gathered = flat.view(G, M, K, dsub + 1)
    total = gathered[0].clone()
    for r in range(1, G):
        total += gathered[r]
    sums.copy_(total[..., :dsub])
    counts.copy_(total[..., dsub].round().to(counts.dtype))
    return sums, counts


def _reseed_empty(C: torch.Tensor, counts: torch.Tensor):
    """Faiss's empty-cluster rule: give an empty centroid a copy of the biggest cluster's centroid, perturbed by
    +-1/1024 (and the donor by the opposite sign)."""
    eps = 1.0 / 1024
    M, K, dsub = C.shape
    empty = (counts == 0).nonzero().cpu().tolist()
    if not empty:
        return 0
    cnt = counts.clone()
    sign = torch.where(torch.arange(dsub, device=C.device) % 2 == 0, 1.0, -1.0) * eps
    for m, k in empty:
        j = int(torch.argmax(cnt[m]))
        C[m, k] = C[m, j] * (1 + sign)
        C[m, j] = C[m, j] * (1 - sign)
        half = cnt[m, j] // 2
        cnt[m, k] = half
        cnt[m, j] -= half
    return len(empty)


def train_pq(x: torch.Tensor, M: int, n_iter: int, centroids: Optional[torch.Tensor] = None, seed: int = SEED):
    """Lloyd k-means of the M sub-quantisers on x [n, D] (device).  Returns (centroids [M,256,dsub], mse)."""
    n, D = x.shape
    dsub = D // M
    if centroids is None:                                   # random-sample initialisation
        perm = torch.from_numpy(np.random.default_rng(seed).permutation(n)[:256].copy()).to(x.device)
        centroids = x[perm].reshape(256, M, dsub).transpose(0, 1).contiguous()
    C = centroids.clone().float().contiguous()
    for it in range(n_iter):
        codes = ops.assign_nearest(x, C, torch.uint8)
        sums, counts = ops.kmeans_stats(x, codes)
        gather_stats_(sums, counts)
        ops.kmeans_update_(sums, counts, C)
        _reseed_empty(C, counts)
    codes = ops.assign_nearest(x, C, torch.uint8)
    mse = float(((ops.decode_raw(codes, C) - x) ** 2).sum(-1).mean())
    return C, mse


def train_opq(x: torch.Tensor, M: int, n_outer: int = 50, n_pq_first: int = 40, n_pq: int = 4, seed: int = SEED):
    """OPQ rotation R [D,D] (x_rot = x @ R) by alternating PQ training and orthogonal Procrustes."""
    n, D = x.shape
    g = torch.Generator(device="cpu").manual_seed(seed)
    R = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0].float().to(x.device)
    C = None
    for it in range(n_outer):
        xr = (x @ R).contiguous()
        C, mse = train_pq(xr, M, n_pq_first if it == 0 else n_pq, centroids=C, seed=seed)
        codes = ops.assign_nearest(xr, C, torch.uint8)
        xrec = ops.decode_raw(codes, C)
        U, _, Vh = torch.linalg.svd((x.T @ xrec).double())       # fp64: keeps R orthogonal to ~1e-7 after the cast
        R = (U @ Vh).float().contiguous()
        if it % 10 == 0 or it == n_outer - 1:
            logger.info("OPQ iteration %d: reconstruction mse %.5f", it, mse)
    return R


def warmup_from_embeds(corpus_embeds: np.ndarray, repconc, opq_iters: int = 50, pq_iters: int = 25,
                       add_chunk: int = 1 << 20):
    """train/run_warmup.py:85-132.  Fills `repconc.rotation` ([d_out, d_in], applied as x @ rotation.T,
    :119-122) and `repconc.centroids` (:124-127), returns (repconc, index) with the whole corpus coded."""
    M, K = repconc.config.MCQ_M, repconc.config.MCQ_K
    assert K == 256, "256 is a standard setting for K. "
    dev = repconc.centroids.device if repconc.centroids.is_cuda else torch.device("cuda", torch.cuda.current_device())
    N, D = corpus_embeds.shape
    take = np.sort(np.random.default_rng(SEED).permutation(N)[:MAX_TRAIN_POINTS])
    xt = torch.from_numpy(np.ascontiguousarray(corpus_embeds[take], dtype=np.float32)).to(dev)
    R = train_opq(xt, M, n_outer=opq_iters) if opq_iters > 0 else torch.eye(D, device=dev)
    C, mse = train_pq((xt @ R).contiguous(), M, pq_iters)
    logger.info("PQ reconstruction mse on the training rows: %.5f", mse)
    with torch.no_grad():
        repconc.rotation.copy_(R.T.to(repconc.rotation.device))        # vt.A is [d_out, d_in]
        repconc.centroids.data.copy_(C.to(repconc.centroids.device))
    if getattr(repconc.config, "similarity_metric", None) == "METRIC_CENTROID_COS":
        repconc.normalize_centrodis()                                   # :129-130
        C = repconc.centroids.data.to(dev)
    index = PQIndex(D, M, 8, METRIC_INNER_PRODUCT, device=dev)
    index.set_centroids(C)
    for i0 in range(0, N, add_chunk):                                   # index.add(corpus_embeds), :114
        chunk = torch.from_numpy(np.ascontiguousarray(corpus_embeds[i0:i0 + add_chunk], dtype=np.float32)).to(dev)
        index.add((chunk @ R).contiguous())
    return repconc, PreTransformIndex(R.T.contiguous(), index)
