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
(`rc_kmeans_update`) are the HIP kernels; the 768-wide GEMMs are library calls on PyTorch-ROCm
and the Procrustes step is a GEMM-only polar iteration (`procrustes_rotation`; the library SVD is its fall-back).  With several ranks each rank passes its corpus shard; everything that feeds the shared model is
combined in rank order so that all ranks hold bit-identical rotations and centroids: the Lloyd statistics
(`gather_stats_`: one all-gather + rank-ordered sum per iteration), the Procrustes matrix x^T x_rec
(`rank_ordered_sum_`), and the random-sample initial centroids (drawn on rank 0, broadcast).  The returned index holds
the local shard's codes with `id_offset` = its global position.
"""
from __future__ import annotations

import functools
import logging
import math
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
    gathered = ops.all_gather(packed, group)             # the handle's IPC / RCCL layer once ops.comm_init() has run
    total = gathered[0].clone()
    for r in range(1, G):
        total += gathered[r]
    sums.copy_(total[..., :dsub])
    counts.copy_(total[..., dsub].round().to(counts.dtype))
    return sums, counts


def _multi(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def rank_ordered_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks with a FIXED order (rank 0 first): all-gather + local adds, so every rank ends with
    the same bits (an all-reduce's ring order is not fixed)."""
    if not _multi(group):
        return t
    G = dist.get_world_size(group)
    parts = ops.all_gather(t.contiguous().view(-1), group)
    total = parts[0].clone()
    for r in range(1, G):
        total += parts[r]
    t.copy_(total.view_as(t))
    return t


def broadcast_from_rank0_(t: torch.Tensor, group=None) -> torch.Tensor:
    if _multi(group):
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return t


def _reseed_empty(C: torch.Tensor, counts: torch.Tensor):
    """Empty clusters after an update, by Faiss's published rule (`split_clusters`, Clustering.cpp of 1.7.x — what
    `index.train` at train/run_warmup.py:113 runs): per sub-quantiser a std::mt19937 seeded with 1234 drives a cyclic walk
    over the clusters that accepts cluster j as the donor with probability (size_j - 1) / (n - 256); the empty cluster gets
    the donor's centroid scaled by (1 +- 1/1024), alternating over the components, the donor the opposite factors, and the
    donor's size is halved for the following draws.  One kernel on the device (rc_kmeans_split_empty): a Lloyd iteration has
    no host synchronisation; every rank holds the same counts, hence makes the same splits."""
    ops.kmeans_split_empty_(C, counts.contiguous())


@functools.lru_cache(maxsize=8)
def _polar_schedule(lower: float):
    """Coefficients (a_k, b_k) of the optimally scaled Newton-Schulz iteration X <- a X + b X X^T X for singular values in
    [lower, 1]: the odd cubic that maps [l, 1] onto [l', 1] with the largest l' equioscillates — p(l) = p(1) = l', maximum 1 at
    x* = sqrt(a / (-3 b)) — which gives, with s = 1 + l + l^2 and m = (2 s / 3) sqrt(s / 3):  a = s / m,  b = -1 / m,
    l' = a l + b l^3  (~2.6 l while l is small; a -> 1.5, b -> -0.5, the plain iteration, as l -> 1).  Known on the host
    without looking at the matrix: the whole iteration enqueues without a synchronisation."""
    out, ell = [], float(lower)
    while 1.0 - ell > 1e-16 and len(out) < 200:
        s = 1.0 + ell + ell * ell
        m = (2.0 * s / 3.0) * math.sqrt(s / 3.0)
        a, b = s / m, -1.0 / m
        out.append((a, b))
        ell = a * ell + b * ell ** 3
    return tuple(out) + ((1.5, -0.5),) * 2                     # two plain steps: quadratic, far below fp64 resolution


def procrustes_rotation(P: torch.Tensor, lower: float = 1e-12, defer: bool = False):
    """argmax_R tr(R^T P) over the orthogonal matrices = U V^T for P = U S V^T — the orthogonal polar factor of P.
    Computed with a Newton-Schulz iteration (GEMMs only: fp64 768^3 products run on the matrix cores in ~35 us each, where
    the Jacobi SVD of the library takes 0.23 s — 50 of them were most of the warm-up).  The plain iteration
    X <- 1.5 X - 0.5 X X^T X multiplies a small singular value by 1.5 per step; the Procrustes matrices of OPQ have
    cond 1e6 ... 1e9, i.e. 45 - 60 steps, and its convergence test is a host synchronisation (rounds 2/3a: 3 - 4 per call,
    17 - 31 ms per OPQ round on a slow host against 7 ms of device time).  Here:
      * X0 = P / |P|_F: every singular value is <= 1 whatever P is (the scaled steps flip the sign of a singular value above
        1 — an estimate of sigma_max is not enough);
      * the steps are scaled optimally for singular values in [lower, 1] (`_polar_schedule`): x2.6 per step while they are
        small, 36 steps for lower = 1e-12, the same for every matrix — nothing is read back while the iteration runs;
      * one orthogonality check at the end: |X^T X - I|_max, on the device.  defer=False reads it (ONE synchronisation) and
        falls back to the library SVD when it is not below 1e-9 (singular P has no unique polar factor; cond above ~1/lower
        does not converge in the fixed number of steps); defer=True returns (X, err) and leaves the decision to the caller."""
    P = P.double()
    n = P.shape[0]
    X = P / torch.linalg.matrix_norm(P)                        # zero / non-finite P: X non-finite, reported by `err`
    for a, b in _polar_schedule(float(lower)):
        X = torch.addmm(X, X, X.T @ X, beta=a, alpha=b)        # one step = two library calls
    err = (X.T @ X - torch.eye(n, dtype=P.dtype, device=P.device)).abs().max()
    if defer:
        return X, err
    if float(err) < 1e-9:                                      # (NaN compares false)
        return X
    U, _, Vh = torch.linalg.svd(P)
    return U @ Vh


def train_pq(x: torch.Tensor, M: int, n_iter: int, centroids: Optional[torch.Tensor] = None, seed: int = SEED,
             mse_on_device: bool = False):
    """Lloyd k-means of the M sub-quantisers on x [n, D] (device).  Returns (centroids [M,256,dsub], mse); with
    `mse_on_device` the mse stays a 0-dim device tensor and the whole call enqueues without a host synchronisation."""
    n, D = x.shape
    dsub = D // M
    if centroids is None:                                   # random-sample initialisation
        perm = torch.from_numpy(np.random.default_rng(seed).permutation(n)[:256].copy()).to(x.device)
        centroids = x[perm].reshape(256, M, dsub).transpose(0, 1).contiguous()
        broadcast_from_rank0_(centroids)                    # several ranks: everyone starts from rank 0's sample
    C = centroids.clone().float().contiguous()
    for it in range(n_iter):
        codes = ops.assign_nearest(x, C, torch.uint8)
        sums, counts = ops.kmeans_stats(x, codes)
        gather_stats_(sums, counts)
        ops.kmeans_update_(sums, counts, C)
        _reseed_empty(C, counts)
    codes = ops.assign_nearest(x, C, torch.uint8)
    err = torch.stack([((ops.decode_raw(codes, C) - x) ** 2).sum().double(),
                       torch.tensor(float(n), dtype=torch.float64, device=x.device)])
    rank_ordered_sum_(err)                                  # MSE over the rows of every rank
    mse = err[0] / err[1]
    return C, (mse if mse_on_device else float(mse))


def _xt_y(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """x^T y in fp64 for two [n, D] fp32 matrices (the Procrustes matrix x^T x_rec).  The library runs the plain product
    [D, n] x [n, D] on 36 output tiles — 36 of 256 CUs, 1.20 ms at 65 536 x 768; as a batch of 16 row slices (16 x 36 tiles)
    whose fp32 partial products are added in fp64, slices ascending, it takes 0.64 ms and is three times closer to the fp64
    product (max error 1.6e-3 against 5.1e-3 on Gaussian rows).  The same order on every rank and in every round."""
    n, D = x.shape
    S = 16
    head = n - n % S
    P = None
    if head:
        P = torch.bmm(x[:head].reshape(S, head // S, D).transpose(1, 2), y[:head].reshape(S, head // S, D)).double().sum(0)
    if head < n:
        tail = (x[head:].T @ y[head:]).double()
        P = tail if P is None else P + tail
    return P


def _procrustes_static(P: torch.Tensor, X: list, Y: torch.Tensor, eye: torch.Tensor, sched: list):
    """procrustes_rotation(P, defer=True) on caller-owned buffers (X: two [D,D] fp64, Y: one): the same library calls in the
    same order, hence the same values; nothing is allocated per step, so the sequence can be captured into a hipGraph."""
    cur, nxt = X
    torch.div(P, torch.linalg.matrix_norm(P), out=cur)
    for a, b in sched:
        torch.mm(cur.T, cur, out=Y)
        nxt.copy_(cur)
        nxt.addmm_(cur, Y, beta=a, alpha=b)
        cur, nxt = nxt, cur
    return cur, (cur.T @ cur - eye).abs().max()


class _RoundGraph:
    """One OPQ round r >= 1 of a single rank as ONE hipGraph on static buffers: rotate (x @ R), n_iter x (assignment,
    statistics, update, empty-cluster rule), the final assignment, decode, squared error, x^T x_rec, the Procrustes
    iteration on its fixed schedule with its orthogonality check, R <- the polar factor, and the round's (mse, check) pair
    written to row `it` of a log that is read once after the last round.  ~70 launches of this package's kernels, ~80 library
    GEMMs and ~40 element-wise ops per round become one graph launch: the host no longer sets the pace (device time 5.1 ms per
    round; an eager round takes 6.7 - 9.7 ms depending on the host).  Round 4 first captured the Lloyd block only, because round
    3 had measured a graph of the Procrustes GEMMs slower than eager calls; measured again (tools: 36 steps, 72 fp64 GEMMs):
    eager 2.07 ms of device time + 0.7 ms of host time, replayed 2.08 ms — identical results.
    `step()` replays the graph, or runs the same body eagerly when the runtime cannot capture it."""

    def __init__(self, x: torch.Tensor, M: int, n_iter: int, n_outer: int, lower: float = 1e-12):
        n, D = x.shape
        dev = x.device
        self.x, self.n, self.n_iter = x, n, n_iter
        self.R = torch.empty((D, D), dtype=torch.float32, device=dev)
        self.C = torch.empty((M, 256, D // M), dtype=torch.float32, device=dev)
        self.xr = torch.empty((n, D), dtype=torch.float32, device=dev)
        self.X = [torch.empty((D, D), dtype=torch.float64, device=dev) for _ in range(2)]
        self.Y = torch.empty((D, D), dtype=torch.float64, device=dev)
        self.eye = torch.eye(D, dtype=torch.float64, device=dev)
        self.log = torch.zeros((n_outer, 2), dtype=torch.float64, device=dev)      # per round: mse, |R^T R - I|_max
        self.it = torch.zeros((1,), dtype=torch.int64, device=dev)                  # the round the next step() is
        self.sched = list(_polar_schedule(float(lower)))
        self.graph = None

    def body(self):
        x, xr, C = self.x, self.xr, self.C
        torch.matmul(x, self.R, out=xr)
        for _ in range(self.n_iter):
            codes = ops.assign_nearest(xr, C, torch.uint8)
            sums, counts = ops.kmeans_stats(xr, codes)
            ops.kmeans_update_(sums, counts, C)
            _reseed_empty(C, counts)
        codes = ops.assign_nearest(xr, C, torch.uint8)
        xrec = ops.decode_raw(codes, C)
        mse = ((xrec - xr) ** 2).sum().double() / float(self.n)
        P = _xt_y(x, xrec)
        cur, err = _procrustes_static(P, self.X, self.Y, self.eye, self.sched)
        self.R.copy_(cur)                                      # fp64 iteration: R stays orthogonal to ~1e-7 after the cast
        self.log.index_copy_(0, self.it, torch.stack([mse, err])[None])
        self.it += 1

    def capture(self):
        side = torch.cuda.Stream(device=self.x.device)
        side.wait_stream(torch.cuda.current_stream(self.x.device))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            self.body()
        self.graph = g

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.body()


def _graph_lloyd_enabled() -> bool:
    import os
    return os.environ.get("RC_WARMUP_GRAPH", "1") != "0"


def train_opq(x: torch.Tensor, M: int, n_outer: int = 50, n_pq_first: int = 40, n_pq: int = 4, seed: int = SEED,
              R0: Optional[torch.Tensor] = None, history: Optional[list] = None, _sync_procrustes: bool = False):
    """OPQ rotation R [D,D] (x_rot = x @ R) by alternating PQ training and orthogonal Procrustes.  `R0`: starting
    rotation (default: the orthogonal polar factor of a seeded Gaussian matrix, the same on every rank); `history`: list
    that receives the reconstruction MSE of every round (what oracle/pq_oracle.py::train_opq returns, for the parity test)."""
    n, D = x.shape
    mses, errs = [], []
    if R0 is None:
        g = torch.Generator(device="cpu").manual_seed(seed)
        # The same seeded Gaussian matrix on every rank, orthogonalised on the device by the iteration the rounds use anyway:
        # the polar factor of a Gaussian matrix is Haar-distributed like the Q of its QR (what Faiss's random rotation
        # takes), costs 2 ms where the library's device QR costs 22 ms (1 300 small kernels) and a host QR 15 - 50 ms, and
        # its orthogonality check joins the rounds' checks.
        G = torch.randn(D, D, generator=g, dtype=torch.float64).to(x.device)
        if _sync_procrustes:
            R0 = procrustes_rotation(G)
        else:
            R0, e0 = procrustes_rotation(G, defer=True)
            errs.append(e0)
    R = R0.float().to(x.device).contiguous()
    C = None
    rg = None                                                   # the rounds >= 1 of a single rank: one hipGraph each
    use_graph = _graph_lloyd_enabled() and not _multi() and n_outer > 2 and n_pq >= 1 and x.is_cuda and not _sync_procrustes
    for it in range(n_outer):
        # No host synchronisation in a round: the Lloyd iterations (assignment, statistics, update, empty-cluster rule), the
        # error and the Procrustes iteration (fixed schedule, orthogonality check left on the device) only enqueue; the
        # checks of all rounds are read once, after the last one.
        if use_graph and it >= 1:
            if rg is None:
                # round 1 runs the round's body eagerly on the static buffers (every lazy initialisation of the libraries has
                # happened by then), rounds >= 2 replay its capture
                rg = _RoundGraph(x, M, n_pq, n_outer)
                rg.R.copy_(R)
                rg.C.copy_(C)
                rg.it.fill_(it)
                rg.body()
                try:
                    rg.capture()
                except Exception as e:                          # a runtime that cannot capture: the same body, eagerly
                    logger.warning("OPQ: hipGraph capture of a round failed (%s); continuing eagerly", e)
            else:
                rg.step()
            continue
        xr = (x @ R).contiguous()
        C, mse = train_pq(xr, M, n_pq_first if it == 0 else n_pq, centroids=C, seed=seed, mse_on_device=True)
        mses.append(mse)
        codes = ops.assign_nearest(xr, C, torch.uint8)
        xrec = ops.decode_raw(codes, C)
        P = _xt_y(x, xrec)
        rank_ordered_sum_(P)                                    # Procrustes matrix over the rows of every rank
        if _sync_procrustes:
            R = procrustes_rotation(P)
        else:
            R, err = procrustes_rotation(P, defer=True)
            errs.append(err)
        R = R.float().contiguous()                              # fp64 iteration: R stays orthogonal to ~1e-7 after the cast
    vals = [float(v) for v in torch.stack(mses + errs).double().cpu()] if mses else []     # the one synchronisation
    mses, errs = vals[:len(mses)], vals[len(mses):]
    if rg is not None:                                          # rounds 1 .. n_outer-1 logged on the device
        rows = rg.log[len(mses):n_outer].cpu()
        mses += [float(v) for v in rows[:, 0]]
        errs += [float(v) for v in rows[:, 1]]
        R = rg.R.clone()
    if any(not (e < 1e-9) for e in errs):
        # a Procrustes matrix the fixed schedule did not orthogonalise (singular, or cond above ~1e12): every rank sees the
        # same values and repeats the training with the checked iteration (library SVD as its fall-back)
        logger.warning("OPQ: a Procrustes step did not converge (max |R^T R - I| = %.3g); repeating with per-round checks",
                       max((e for e in errs if e == e), default=float("nan")))
        return train_opq(x, M, n_outer, n_pq_first, n_pq, seed, R0, history, _sync_procrustes=True)
    if history is not None:
        history.extend(mses)
    for it, v in enumerate(mses):
        if it % 10 == 0 or it == n_outer - 1:
            logger.info("OPQ iteration %d: reconstruction mse %.5f", it, v)
    return R


def warmup_from_embeds(corpus_embeds: np.ndarray, repconc, opq_iters: int = 50, pq_iters: int = 25,
                       add_chunk: int = 1 << 20):
    """train/run_warmup.py:85-132.  Fills `repconc.rotation` ([d_out, d_in], applied as x @ rotation.T,
    :119-122) and `repconc.centroids` (:124-127), returns (repconc, index) with the whole corpus coded."""
    M, K = repconc.config.MCQ_M, repconc.config.MCQ_K
    assert K == 256, "256 is a standard setting for K. "
    dev = repconc.centroids.device if repconc.centroids.is_cuda else torch.device("cuda", torch.cuda.current_device())
    N, D = corpus_embeds.shape
    if _multi():
        # corpus sharded over the ranks (BASELINE configs[2]): the per-shard statistics of every Lloyd iteration and
        # the Procrustes matrices travel through the handle's own exchange layer (csrc/comm.hip: IPC peer stores over
        # xGMI, or RCCL) — torch.distributed only carries the one-time handshake
        # comm_init is collective: all ranks end on the same transport (ipc, else rccl) or ALL raise — then all of them use
        # torch.distributed's all-gathers (ops.all_gather picks its path from a state every rank shares)
        try:
            with torch.cuda.device(dev):
                ops.comm_init()
        except Exception as e:
            logger.warning("native exchange layer unavailable (%s); using torch.distributed all-gathers", e)
    take = np.sort(np.random.default_rng(SEED).permutation(N)[:MAX_TRAIN_POINTS])
    xt = torch.from_numpy(np.ascontiguousarray(corpus_embeds[take], dtype=np.float32)).to(dev)
    R = train_opq(xt, M, n_outer=opq_iters) if opq_iters > 0 else torch.eye(D, device=dev)
    C, mse = train_pq((xt @ R).contiguous(), M, pq_iters)
    if _multi():
        with torch.cuda.device(dev):
            ops.comm_check()                                            # a timed-out statistics gather: stop, do not save
    logger.info("PQ reconstruction mse on the training rows: %.5f", mse)
    with torch.no_grad():
        repconc.rotation.copy_(R.T.to(repconc.rotation.device))        # vt.A is [d_out, d_in]
        repconc.centroids.data.copy_(C.to(repconc.centroids.device))
    if getattr(repconc.config, "similarity_metric", None) == "METRIC_CENTROID_COS":
        repconc.normalize_centrodis()                                   # :129-130
        C = repconc.centroids.data.to(dev)
    index = PQIndex(D, M, 8, METRIC_INNER_PRODUCT, device=dev)
    index.set_centroids(C)
    if _multi():                                                        # this rank's shard starts after the lower ranks' rows
        sizes = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=dev)
        sizes[dist.get_rank()] = N
        rank_ordered_sum_(sizes)
        index.id_offset = int(sizes[: dist.get_rank()].sum())
    for i0 in range(0, N, add_chunk):                                   # index.add(corpus_embeds), :114
        chunk = torch.from_numpy(np.ascontiguousarray(corpus_embeds[i0:i0 + add_chunk], dtype=np.float32)).to(dev)
        index.add((chunk @ R).contiguous())
    return repconc, PreTransformIndex(R.T.contiguous(), index)


# ---------------------------------------------------------------------------------------------------------------- CLI
def main(argv=None):
    """`python -m repconc_amd.train.run_warmup --model_name_or_path ... --MCQ_M 48 --input_corpus_embed_path ...`
    with the arguments of the reference's script (train/run_warmup.py:22-83,135-189): builds the RepCONC model around
    the dense encoder, fits rotation + centroids on the corpus embeddings, saves the model, the tokenizer, the IndexPQ
    file over the rotated vectors and the corpus ids."""
    import os
    from dataclasses import dataclass, field
    from transformers import AutoConfig, AutoTokenizer, HfArgumentParser, set_seed
    from ..faiss_io import write_index
    from ..models.dense import AutoDense
    from ..models.repconc import RepCONC

    @dataclass
    class DataArguments:
        input_corpus_embed_path: str = field(metadata={"help": "corpus embeddings (.npy)"})
        input_corpus_ids_path: str = field(metadata={"help": "corpus ids (.npy), copied next to the index"})
        output_model_dir: str = field(metadata={"help": "where to save the RepCONC model"})
        output_index_path: str = field(metadata={"help": "where to save the index"})
        output_corpus_ids_path: str = field(metadata={"help": "where to save the corpus ids"})

    @dataclass
    class ModelArguments:
        model_name_or_path: str = field(metadata={"help": "path of the dense encoder"})
        MCQ_M: int = field(metadata={"help": "number of sub-vectors per text"})
        similarity_metric: str = field(default=None, metadata={"choices": ["METRIC_CENTROID_COS", "METRIC_IP", "METRIC_COS"]})
        pooling: str = field(default=None, metadata={"choices": ["cls", "mean"]})
        MCQ_K: int = field(default=256)

    model_args, data_args = HfArgumentParser((ModelArguments, DataArguments)).parse_args_into_dataclasses(argv)
    logging.basicConfig(format="%(asctime)s-%(levelname)s-%(name)s- %(message)s", level=logging.INFO)
    set_seed(2022)
    config = AutoConfig.from_pretrained(model_args.model_name_or_path)
    config.MCQ_M, config.MCQ_K = model_args.MCQ_M, model_args.MCQ_K
    if model_args.similarity_metric is not None:
        config.similarity_metric = model_args.similarity_metric
    if model_args.pooling is not None:
        config.pooling = model_args.pooling
    tokenizer = AutoTokenizer.from_pretrained(model_args.model_name_or_path)
    encoder = AutoDense.from_pretrained(model_args.model_name_or_path, config=config)
    repconc = RepCONC(config, encoder, use_constraint=False, sk_epsilon=None, sk_iters=None).to("cuda")
    repconc, index = warmup_from_embeds(np.load(data_args.input_corpus_embed_path), repconc)
    os.makedirs(data_args.output_model_dir, exist_ok=True)
    repconc.save_pretrained(data_args.output_model_dir)
    tokenizer.save_pretrained(data_args.output_model_dir)
    for path in (data_args.output_index_path, data_args.output_corpus_ids_path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    write_index(index.index, data_args.output_index_path)
    np.save(data_args.output_corpus_ids_path, np.load(data_args.input_corpus_ids_path))


if __name__ == "__main__":
    main()
