"""numpy restatement of the RepCONC PQ hot path.  TEST INFRASTRUCTURE ONLY.

Parity status
-------------
* rows a-1 … a-8 of SURVEY.md §8 (distance table, centring, Sinkhorn, argmax/argmin,
  decode, diagnostics): PINNED.  tests/test_oracle_golden.py checks every function here
  against tests/golden/*.npz, which oracle/gen_golden.py produced by importing the
  reference (`/root/reference/src/repconc/models/repconc/modeling_repconc.py`) in the
  build container and running it on torch-CPU.
* rows a-9 … a-12 (Faiss IndexPQ add/search, k-means warm-up): PARITY UNPINNED.  Faiss is
  not vendored in the reference and not installed anywhere we can reach (no network); the
  restatement follows Faiss 1.7.x's published behaviour and is
  anchored on the reference's call sites plus the identity
  ``sum_m LUT[m][code_m] == <q, decode(code)>`` that the reference's own `decode` pins.

Every function cites the reference lines it restates (paths relative to
/root/reference/src/repconc/).  Nothing here is imported by the product package.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64


# --------------------------------------------------------------------------- a-1
def _multi_row_sum(rows):
    """torch's cascade (aten/native/cpu/SumKernel.cpp `multi_row_sum`, 4 levels, level step 16): `rows` is the list of
    per-round operands (each an array holding the 4 ILP partials side by side); after every 16 rounds the running sum moves
    one level up, the levels are merged lowest first at the end.  Up to 16 rounds this is a plain running sum from 0."""
    num_levels, power = 4, 4
    step, mask = 1 << power, (1 << power) - 1
    acc = [np.zeros_like(rows[0]) if rows else None for _ in range(num_levels)]
    if not rows:
        return None
    size, i = len(rows), 0
    while i + step <= size:
        for _ in range(step):
            acc[0] = acc[0] + rows[i]
            i += 1
        for j in range(1, num_levels):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = np.zeros_like(acc[j - 1])
            if i & (mask << (j * power)):
                break
    while i < size:
        acc[0] = acc[0] + rows[i]
        i += 1
    for j in range(1, num_levels):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def _rowsum_torch_cpu_order(sq: np.ndarray) -> np.ndarray:
    """Sum over the last axis in the order torch-CPU's `sum(-1)` uses for a contiguous fp32 row of length dsub
    (aten/native/cpu/SumKernel.cpp; bit-verified against torch 2.10 for every divisor of 768).
    dsub >= 8 (`vectorized_inner_sum`): split the row into 8-wide vectors v0,v1,…; four ILP accumulators acc_j = sum_i
    v_{4i+j} (a cascade: every 16 rounds — 512 floats — the running sums move to a second level, so dsub = 768 is NOT a
    plain running sum); left-over vectors go to acc_0 in order; acc_0 += acc_1, acc_2, acc_3; result = ((tail scalars
    summed from 0) + lane0) + lane1 … + lane7.
    dsub < 8 (`scalar_inner_sum`): the same with scalars for vectors: p_j = x_j (j < 4) when dsub >= 4, the remaining
    scalars are added to p_0 in order, then p_0 += p_1, p_2, p_3."""
    dsub = sq.shape[-1]
    lead = sq.shape[:-1]
    if dsub < 8:
        p = [np.zeros(lead, F32) for _ in range(4)]
        full = dsub // 4
        if full:
            for j in range(4):
                p[j] = p[j] + sq[..., j]
        for j in range(4 * full, dsub):
            p[0] = p[0] + sq[..., j]
        r = p[0] + p[1]
        r = r + p[2]
        return r + p[3]
    nv, tail = dsub // 8, dsub % 8
    full = nv // 4
    rounds = [sq[..., 32 * i: 32 * i + 32] for i in range(full)]          # [.., 4 x 8]: the four ILP partials of a round
    acc = _multi_row_sum(rounds)
    acc = np.zeros(lead + (32,), F32) if acc is None else acc
    a0 = acc[..., 0:8]
    for v in range(4 * full, nv):
        a0 = a0 + sq[..., 8 * v: 8 * v + 8]
    a = a0 + acc[..., 8:16]
    a = a + acc[..., 16:24]
    a = a + acc[..., 24:32]
    r = np.zeros(lead, F32)
    for j in range(tail):
        r = r + sq[..., nv * 8 + j]
    for lane in range(8):
        r = r + a[..., lane]
    return r


def dist_table(x: np.ndarray, centroids: np.ndarray, chunk: int = 256) -> np.ndarray:
    """d[m,b,k] = sum_j (x[b, m*dsub+j] - C[m,k,j])**2, fp32, sub→round, square→round,
    torch-CPU sum order.  models/repconc/modeling_repconc.py:49-50."""
    x = np.ascontiguousarray(x, dtype=F32)
    C = np.ascontiguousarray(centroids, dtype=F32)
    B, D = x.shape
    M, K, dsub = C.shape
    assert D == M * dsub
    out = np.empty((M, B, K), F32)
    xs = x.reshape(B, M, dsub)
    for b0 in range(0, B, chunk):
        xb = xs[b0:b0 + chunk].transpose(1, 0, 2)[:, :, None, :]      # [M,bc,1,dsub]
        diff = xb - C[:, None, :, :]                                    # [M,bc,K,dsub]
        out[:, b0:b0 + chunk, :] = _rowsum_torch_cpu_order(diff * diff)
    return out


# --------------------------------------------------------------------------- a-2
def minmax_per_m(d: np.ndarray):
    """per-m max / min over (b,k).  modeling_repconc.py:76-77."""
    return d.max(axis=(1, 2)).astype(F32), d.min(axis=(1, 2)).astype(F32)


def centre(d: np.ndarray, mx: np.ndarray, mn: np.ndarray) -> np.ndarray:
    """(d - mid)/amp with mid=(mx+mn)/2, amp=(mx-mid)+1e-5, all fp32, IEEE division.
    modeling_repconc.py:81-84 (mx/mn are the already all-reduced values of :78-80)."""
    mx = mx.astype(F32)
    mn = mn.astype(F32)
    mid = (mx + mn) / F32(2)
    amp = (mx - mid) + F32(1e-5)
    assert np.all(amp > 0)                                              # :83
    return ((d - mid[:, None, None]) / amp[:, None, None]).astype(F32)


# --------------------------------------------------------------------------- a-3
def sinkhorn_q(out_shards, epsilon: float, iters: int):
    """fp64 Sinkhorn-Knopp in the reference's in-place Q form.  modeling_repconc.py:137-165.

    `out_shards` is a list with one [M,K,B_r] fp64 array per simulated rank (a single
    element = not distributed).  Cross-rank all_reduce(SUM) of :151 and :157 is restated
    as a rank-ordered sum.  Returns the list of per-rank Q."""
    G = len(out_shards)
    Qs = [np.exp(o / epsilon) for o in out_shards]                       # :141
    K = Qs[0].shape[1]
    B = Qs[0].shape[2] * G                                              # :144,:150
    tot = sum(q.sum(-1, keepdims=True).sum(-2, keepdims=True) for q in Qs)   # :148-151
    for q in Qs:
        q /= tot                                                        # :152
    for _ in range(iters):                                              # :153
        rows = sum(q.sum(axis=2, keepdims=True) for q in Qs)            # :155-157
        for q in Qs:
            q /= rows                                                   # :158
            q /= K                                                      # :159
            q /= q.sum(axis=1, keepdims=True)                           # :162
            q /= B                                                      # :163
    for q in Qs:
        q *= B                                                          # :164
    return Qs


# ----------------------------------------------------------------- a-4 / a-5 (quantize)
def quantize(x, centroids, use_constraint: bool, epsilon: float = 0.003, iters: int = 100,
             shards: int = 1, return_intermediates: bool = False):
    """codes int64 [B,M].  modeling_repconc.py:47-67.  `shards`>1 simulates the
    dist.is_initialized() branch with equal row blocks per rank (:78-80,:149-157)."""
    x = np.ascontiguousarray(x, dtype=F32)
    B = x.shape[0]
    d = dist_table(x, centroids)
    if not use_constraint:
        codes = np.argmin(d, axis=-1)                                   # :52 first-min
        return (codes.T.copy(), {"dist": d}) if return_intermediates else codes.T.copy()
    mx, mn = minmax_per_m(d)                                            # global == max over ranks
    dc = centre(d, mx, mn)                                              # :54
    assert B % shards == 0
    bl = B // shards
    outs = [-(dc[:, r * bl:(r + 1) * bl, :].astype(F64)).transpose(0, 2, 1) for r in range(shards)]
    Qs = sinkhorn_q(outs, epsilon, iters)                               # :57-62
    Q = np.concatenate([q.transpose(0, 2, 1) for q in Qs], axis=1)     # M,B,K
    codes = np.argmax(Q, axis=-1)                                       # :63 first-max
    flags = int(np.isnan(Q).any()) | (int(np.isinf(Q).any()) << 1)      # :64-65
    codes = codes.T.copy()                                              # :66
    if return_intermediates:
        return codes, {"dist": d, "mx": mx, "mn": mn, "centred": dc, "flags": flags, "Q": Q}
    return codes


def codes_equal_up_to_fp64_ties(got: np.ndarray, want: np.ndarray, Q: np.ndarray, rtol: float = 1e-9):
    """True if `got` equals the reference codes `want` except where the reference's own transport
    plan Q [M,B,K] holds an fp64-rounding-level tie: Q[m,b,got] >= (1-rtol) * Q[m,b,want].

    Such ties are structural when B <~ K or after very few iterations: a centroid whose row sum is
    dominated by ONE document gets Q_kb/rowsum_k == 1.0 exactly (modeling_repconc.py:158), so whole
    groups of entries tie and the reference's argmax is decided by the last ulp of its exp() — the
    numpy and C restatements of the reference already disagree with each other there.  Returns
    (ok, n_mismatch)."""
    got = np.asarray(got).astype(np.int64)
    want = np.asarray(want).astype(np.int64)
    bad = np.argwhere(got != want)
    ok = True
    for b, m in bad:
        if not Q[m, b, got[b, m]] >= (1.0 - rtol) * Q[m, b, want[b, m]]:
            ok = False
    return ok, len(bad)


def sinkhorn_codes_logdomain(dc: np.ndarray, epsilon: float, iters: int) -> np.ndarray:
    """Same codes as quantize(use_constraint=True) computed with potentials instead of the
    in-place matrix (the formulation the HIP kernels use).  Used by the
    tests as an independent cross-check of the algebra, not as the parity oracle."""
    L = -(dc.astype(F64)) / epsilon                                     # [M,B,K]
    M, B, K = L.shape
    f = -np.log(np.exp(L).sum(axis=1))                                  # pass 0, g=0  [M,K]
    g = np.zeros((M, B), F64)
    for _ in range(iters - 1):
        w = np.exp(L + f[:, None, :] + g[:, :, None])
        c = w.sum(axis=2)
        g = g - np.log(c)
        f = f - np.log((w / c[:, :, None]).sum(axis=1))
    return np.argmax(L + f[:, None, :], axis=-1).T.copy()


# --------------------------------------------------------------------------- a-6
def decode(codes: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """out[b, m*dsub:(m+1)*dsub] = C[m, codes[b,m], :].  modeling_repconc.py:168-184."""
    M, K, dsub = centroids.shape
    n = codes.shape[0]
    idx = codes.astype(np.int64)
    return centroids[np.arange(M)[None, :], idx, :].reshape(n, M * dsub)


def decode_bwd(codes: np.ndarray, grad_out: np.ndarray, M: int, K: int) -> np.ndarray:
    """Gradient of decode w.r.t. the centroids: scatter-add of grad_out rows into
    [M,K,dsub] (autograd of the advanced-index gather at modeling_repconc.py:175)."""
    n, D = grad_out.shape
    dsub = D // M
    g = np.zeros((M, K, dsub), F32)
    go = grad_out.reshape(n, M, dsub)
    for m in range(M):
        np.add.at(g[m], codes[:, m].astype(np.int64), go[:, m, :])
    return g


# --------------------------------------------------------------------------- a-8
def normalize_centroids(centroids: np.ndarray) -> np.ndarray:
    """C / max(||C||_2, 1e-12) over dsub (F.normalize).  modeling_repconc.py:112-116."""
    n = np.sqrt((centroids.astype(F32) ** 2).sum(-1, keepdims=True)).astype(F32)
    return (centroids / np.maximum(n, F32(1e-12))).astype(F32)


# --------------------------------------------------------------------------- a-13
def code_histogram(codes: np.ndarray, K: int = 256) -> np.ndarray:
    """hist[m,k] = #{b : codes[b,m]==k} (the 256 `.sum().item()` calls of
    models/repconc/finetune_repconc.py:590-592, for every sub-quantiser at once)."""
    B, M = codes.shape
    h = np.zeros((M, K), np.int32)
    for m in range(M):
        h[m] = np.bincount(codes[:, m].astype(np.int64), minlength=K)
    return h


def eval_balance(codes: np.ndarray, block_id: int = 0):
    """finetune_repconc.py:580-597."""
    col = codes[:, block_id]
    n = len(col)
    bal = [abs(1 - int((col == i).sum()) / (n / 256)) for i in range(256)]
    return {"avg_imbalance": round(float(np.mean(bal)), 3),
            "max_imbalance": round(float(np.max(bal)), 3)}


def test_quantize(x, centroids, epsilon, iters, block_id: int = 0):
    """finetune_repconc.py:600-613 (note: sqrt of the summed squares, then mean)."""
    out = {}
    for prefix, uc in (("w/o_conc", False), ("w/_conc", True)):
        codes = quantize(x, centroids, uc, epsilon, iters)
        q = decode(codes, centroids)
        mse = np.sqrt(((q - x) ** 2).sum(-1)).mean()
        out[f"{prefix}_mse"] = round(float(mse), 3)
        out.update({f"{prefix}_{k}": v for k, v in eval_balance(codes, block_id).items()})
    return out


# ----------------------------------------------------------------- a-9 … a-11 (Faiss side)
def adc_lut(q: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """Inner-product look-up tables LUT[nq,M,K] = <q_m, C[m,k]> (Faiss IndexPQ search with
    METRIC_INNER_PRODUCT, called at models/repconc/evaluate_repconc.py:182).  fp32, the dot
    product accumulated j-ascending like Faiss's scalar `fvec_inner_product`.
    PARITY UNPINNED (no Faiss to run)."""
    nq, D = q.shape
    M, K, dsub = centroids.shape
    qs = q.reshape(nq, M, dsub).astype(F32)
    lut = np.zeros((nq, M, K), F32)
    for j in range(dsub):
        lut = lut + qs[:, :, None, j] * centroids[None, :, :, j]
    return lut


def adc_scores(lut: np.ndarray, codes: np.ndarray) -> np.ndarray:
    """score[q,n] = sum_m LUT[q,m,codes[n,m]], m-ascending fp32 accumulation from 0."""
    nq, M, K = lut.shape
    s = np.zeros((nq, codes.shape[0]), F32)
    for m in range(M):
        s = s + lut[:, m, :][:, codes[:, m].astype(np.int64)]
    return s


def topk_desc(scores: np.ndarray, k: int):
    """Top-k per row, ordered (score descending, index ascending) — the deterministic
    refinement of Faiss's 'results sorted by decreasing inner product'."""
    nq, n = scores.shape
    k = min(k, n)
    order = np.lexsort((np.broadcast_to(np.arange(n), scores.shape), -scores.astype(F64)), axis=1)
    idx = order[:, :k]
    return np.take_along_axis(scores, idx, axis=1), idx.astype(np.int64)


def adc_search(q, centroids, codes, k):
    """evaluate_repconc.py:180-185 with a Faiss IndexPQ(IP) — brute force."""
    return topk_desc(adc_scores(adc_lut(q, centroids), codes), k)


# --------------------------------------------------------------------------- a-12
def kmeans_stats(x: np.ndarray, codes: np.ndarray, M: int, K: int = 256):
    """Per-(m,k) sufficient statistics of one Lloyd step: sum[M,K,dsub] fp64 (so the value
    is order-independent to ~1e-16) and count[M,K].  Stands for the centroid-update half of
    Faiss's k-means inside `index.train` (train/run_warmup.py:113)."""
    B, D = x.shape
    dsub = D // M
    sums = np.zeros((M, K, dsub), F64)
    cnt = np.zeros((M, K), np.int64)
    xs = x.reshape(B, M, dsub).astype(F64)
    for m in range(M):
        np.add.at(sums[m], codes[:, m].astype(np.int64), xs[:, m, :])
        cnt[m] = np.bincount(codes[:, m].astype(np.int64), minlength=K)
    return sums, cnt


def kmeans_update(sums, cnt, old_centroids):
    """new C = sum/count where count>0, else keep the old centroid."""
    new = old_centroids.astype(F32).copy()
    nz = cnt > 0
    new[nz] = (sums[nz] / cnt[nz][:, None]).astype(F32)
    return new


def lloyd(x, centroids, n_iter: int):
    C = centroids.astype(F32).copy()
    M, K, _ = C.shape
    for _ in range(n_iter):
        codes = quantize(x, C, False)
        s, c = kmeans_stats(x, codes, M, K)
        C = kmeans_update(s, c, C)
    return C


def reseed_empty(C, cnt, n=None):
    """Faiss 1.7.x `split_clusters` (Clustering.cpp; restated from the published source), applied to
    every sub-quantiser as Faiss's per-sub-quantiser `Clustering` objects do: with a generator re-seeded to 1234
    (std::mt19937; numpy's legacy RandomState produces the same stream) walk the clusters cyclically from 0 and accept
    cluster cj as the donor with probability (size_cj - 1) / (n - k) — a size-proportional draw, not the arg-max; the empty
    cluster takes a copy of the donor's centroid, the copy scaled by (1 +- 1/1024) with alternating sign over the components
    and the donor by the opposite factor; the donor's (float) size is halved.  `n`: number of training points (default: the
    sum of the counts).  Returns the number of splits."""
    eps = F32(1.0 / 1024)
    M, K, dsub = C.shape
    nsplit = 0
    for m in range(M):
        h = cnt[m].astype(F32)
        if not (h == 0).any():
            continue
        nn = int(cnt[m].sum()) if n is None else int(n)
        denom = np.float64(F32(nn - K))
        rs = np.random.RandomState(1234)
        raw = rs._bit_generator.random_raw
        for ci in range(K):
            if h[ci] != 0:
                continue
            cj, draws = 0, 0
            if denom <= 0 or h.max() <= 1:                   # Faiss would spin forever: take the biggest cluster
                cj = int(np.argmax(h))
            else:
                while True:
                    p = F32((np.float64(h[cj]) - 1.0) / denom)
                    r = F32(raw()) / F32(4294967295.0)
                    draws += 1
                    if r < p or draws > 10_000_000:
                        break
                    cj = (cj + 1) % K
            C[m, ci] = C[m, cj]
            even = np.arange(dsub) % 2 == 0
            C[m, ci] = np.where(even, C[m, ci] * (F32(1) + eps), C[m, ci] * (F32(1) - eps)).astype(F32)
            C[m, cj] = np.where(even, C[m, cj] * (F32(1) - eps), C[m, cj] * (F32(1) + eps)).astype(F32)
            h[ci] = h[cj] / F32(2)
            h[cj] = h[cj] - h[ci]
            nsplit += 1
    return nsplit


def train_pq(x, M, n_iter, centroids=None, seed=1234, quantize_fn=None):
    """The PQ half of `index.train` (train/run_warmup.py:113): Lloyd iterations of the M sub-quantisers on x [n, D]
    from a random sample of 256 training rows (numpy default_rng(seed) permutation — the build's choice; Faiss draws
    with its own generator, which cannot be reproduced here) or from `centroids`; returns (C, mse) with
    mse = mean_b ||decode(code_b) - x_b||^2 after a final assignment.  `quantize_fn` (default: this file's exact-order
    `quantize`) lets a test substitute the compiled C restatement of the same arithmetic for speed."""
    q = quantize_fn or (lambda xx, cc: quantize(xx, cc, False))
    n, D = x.shape
    dsub = D // M
    if centroids is None:
        perm = np.random.default_rng(seed).permutation(n)[:256]
        centroids = np.ascontiguousarray(x[perm].reshape(256, M, dsub).transpose(1, 0, 2))
    C = centroids.astype(F32).copy()
    for _ in range(n_iter):
        codes = q(x, C)
        s, c = kmeans_stats(x, codes, M)
        C = kmeans_update(s, c, C)
        reseed_empty(C, c)
    codes = q(x, C)
    rec = decode(codes, C)
    mse = float(((rec.astype(F64) - x.astype(F64)) ** 2).sum() / n)
    return C, mse


def train_opq(x, M, R0, n_outer, n_pq_first, n_pq, seed=1234, quantize_fn=None):
    """The OPQ half of `index.train` (the `OPQ{M}` pre-transform of run_warmup.py:92-96), non-parametric OPQ as Faiss's
    OPQMatrix::train runs it (Ge et al. 2013): from the orthogonal R0, n_outer rounds of { x_r = x R ; PQ k-means on x_r
    (n_pq_first Lloyd iterations the first round, n_pq warm-started ones afterwards) ; x_rec = decode(encode(x_r)) ;
    R = U V^T with x^T x_rec = U S V^T }.  Returns (R, [mse per round])."""
    q = quantize_fn or (lambda xx, cc: quantize(xx, cc, False))
    R = R0.astype(F32).copy()
    C, hist = None, []
    for it in range(n_outer):
        xr = np.ascontiguousarray((x.astype(F32) @ R).astype(F32))
        C, mse = train_pq(xr, M, n_pq_first if it == 0 else n_pq, centroids=C, seed=seed, quantize_fn=quantize_fn)
        hist.append(mse)
        xrec = decode(q(xr, C), C)
        P = x.astype(F64).T @ xrec.astype(F64)
        U, _, Vh = np.linalg.svd(P)
        R = (U @ Vh).astype(F32)
    return R, hist


# --------------------------------------------------------------------------- metric
def mrr_at_k(ranked_ids: np.ndarray, positives, k: int = 10) -> float:
    """MRR@k as utils/eval_utils.py:136-190 computes it through pytrec_eval: truncate each
    run to its top-k, reciprocal rank of the first doc with relevance>=1, mean over
    queries, rounded to 5 dp.  `positives[i]` is the set of relevant ids for query i."""
    rr = []
    for row, pos in zip(ranked_ids, positives):
        r = 0.0
        for rank, did in enumerate(row[:k]):
            if int(did) in pos:
                r = 1.0 / (rank + 1)
                break
        rr.append(r)
    return round(float(np.mean(rr)), 5)


# --------------------------------------------------------------------------- IVF extension
def ivf_search(q, centroids, codes, list_ids, coarse, k, nprobe):
    """Brute-force restatement of repconc_amd.ivf.IVFPQIndex.search (a build-side extension with no reference
    counterpart): probe the nprobe cells with the largest <q, coarse centroid> (ties: lower cell),
    score the rows of those cells with the flat ADC arithmetic, top-k by (score desc, id asc); -1 / -inf padding."""
    nq = q.shape[0]
    cs = q.astype(F32) @ coarse.astype(F32).T
    order = np.lexsort((np.broadcast_to(np.arange(cs.shape[1]), cs.shape), -cs.astype(F64)), axis=1)[:, :nprobe]
    lut = adc_lut(q, centroids)
    out_s = np.full((nq, k), -np.inf, F32)
    out_i = np.full((nq, k), -1, np.int64)
    for qi in range(nq):
        rows = np.nonzero(np.isin(list_ids, order[qi]))[0]
        if len(rows) == 0:
            continue
        s = adc_scores(lut[qi:qi + 1], codes[rows])[0]
        o = np.lexsort((rows, -s.astype(F64)))[:k]
        out_s[qi, :len(o)] = s[o]
        out_i[qi, :len(o)] = rows[o]
    return out_s, out_i
