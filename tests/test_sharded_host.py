"""The N>1 path on CPU: repconc_amd.sharded's collective choreography under gloo (world_size 2)
and as virtual shards, with the numpy stage stand-in (tests/sharded_stub.py).  The golden
fixtures say what the reference's own gloo run produced (shard{2,4}_equal == 1)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_case

EPS, ITERS = 0.003, 100
CASE = "m8_b300_gauss"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret, split):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from repconc_amd.sharded import TorchDistComm, assign_sinkhorn_sharded
    from sharded_stub import NumpyStages
    _, x, C = load_case(CASE)
    bl = x.shape[0] // world
    codes, flags = assign_sinkhorn_sharded(torch.from_numpy(x[rank * bl:(rank + 1) * bl]), torch.from_numpy(C),
                                           EPS, ITERS, TorchDistComm(), stages=NumpyStages(), split=split)
    ret[rank] = codes.numpy().astype(np.uint8)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,split", [(2, False), (2, True)])
def test_gloo_sharded_equals_reference_codes(world, split):
    """split=True is the multi-GPU default: two halves of M in lock-step with ASYNC all-gathers (gloo here)."""
    g, x, C = load_case(CASE)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret, split), nprocs=world, join=True)
        got = np.concatenate([ret[r] for r in range(world)], 0)
    assert np.array_equal(got, g["codes_constrained"][: got.shape[0]] if got.shape[0] != x.shape[0]
                          else g["codes_constrained"])


@pytest.mark.parametrize("shards", [1, 3, 5])
def test_virtual_shards_equal_reference_codes(shards):
    from repconc_amd.sharded import assign_sinkhorn_virtual
    from sharded_stub import NumpyStages
    g, x, C = load_case(CASE)
    bl = x.shape[0] // shards
    xs = [torch.from_numpy(x[r * bl:(r + 1) * bl]) for r in range(shards)]
    for split in (False, True):
        codes, flags = assign_sinkhorn_virtual(xs, torch.from_numpy(C), EPS, ITERS, stages=NumpyStages(), split=split)
        got = torch.cat(codes, 0).numpy().astype(np.uint8)
        assert np.array_equal(got, g["codes_constrained"])


def test_single_row_global_batch_is_all_zero_codes():
    """B_global == 1: the reference's first row normalisation makes an exact K-way tie."""
    from repconc_amd.sharded import SingleComm, assign_sinkhorn_sharded
    from sharded_stub import NumpyStages
    from oracle import pq_oracle
    _, x, C = load_case(CASE)
    want = pq_oracle.quantize(x[:1], C, True, EPS, ITERS)
    assert not want.any()
    codes, _ = assign_sinkhorn_sharded(torch.from_numpy(x[:1]), torch.from_numpy(C), EPS, ITERS, SingleComm(),
                                       stages=NumpyStages())
    assert np.array_equal(codes.numpy(), want)


def _stats_worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pq_oracle
    from repconc_amd.train.run_warmup import gather_stats_
    _, x, C = load_case(CASE)
    n = x.shape[0] // world
    xs = x[rank * n:(rank + 1) * n]
    codes = pq_oracle.quantize(xs, C, False)
    sums, counts = pq_oracle.kmeans_stats(xs, codes, C.shape[0])
    s, c = torch.from_numpy(sums.astype(np.float64)), torch.from_numpy(counts.astype(np.int64))
    gather_stats_(s, c)
    ret[rank] = (s.numpy().copy(), c.numpy().copy())
    dist.destroy_process_group()


def test_gloo_kmeans_statistics_all_gather_is_rank_identical():
    """Index-build sharding: per-shard Lloyd statistics -> one all-gather -> rank-ordered sum.  Both ranks
    must hold bit-identical totals, equal to the unsharded statistics (counts exactly, sums to fp64 rounding)."""
    from oracle import pq_oracle
    _, x, C = load_case(CASE)
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_stats_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        (s0, c0), (s1, c1) = ret[0], ret[1]
    assert np.array_equal(s0.view(np.uint64), s1.view(np.uint64)) and np.array_equal(c0, c1)
    n = x.shape[0] // world * world
    codes = pq_oracle.quantize(x[:n], C, False)
    sums, counts = pq_oracle.kmeans_stats(x[:n], codes, C.shape[0])
    assert np.array_equal(c0, counts)
    np.testing.assert_allclose(s0, sums, rtol=1e-12, atol=1e-12)


def _loss_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from repconc_amd.train.stage1 import Stage1Config, global_loss_and_local_grads
    data = _loss_case()
    nq, np_, nn_ = data["q"].shape[0] // world, data["pos"].shape[0] // world, data["neg"].shape[0] // world
    sl = lambda t, n: t[rank * n:(rank + 1) * n]
    loss, gq, gp, gn = global_loss_and_local_grads(
        sl(data["q"], nq), sl(data["pos"], np_), sl(data["neg"], nn_), sl(data["qids"], nq), sl(data["pos_ids"], np_),
        sl(data["neg_ids"], nn_), data["qrels"], Stage1Config(dynamic_topk_hard_negative=5), "METRIC_IP", 8)
    ret[rank] = (float(loss), gq.numpy().copy(), gp.numpy().copy(), gn.numpy().copy())
    dist.destroy_process_group()


def _loss_case():
    g = torch.Generator().manual_seed(5)
    nq, nn_ = 8, 16
    return {"q": torch.randn(nq, 32, generator=g), "pos": torch.randn(nq, 32, generator=g),
            "neg": torch.randn(nn_, 32, generator=g), "qids": torch.arange(nq), "pos_ids": torch.arange(100, 100 + nq),
            "neg_ids": torch.tensor([100, 201, 202, 203, 204, 205, 206, 207, 208, 209, 210, 211, 212, 213, 214, 103]),
            "qrels": {i: [100 + i, 300 + i] for i in range(nq)} | {0: [100, 201]}}


def test_gloo_stage1_global_loss_local_gradient_slices():
    """N2: two ranks each hold half of the queries / positives / negatives; the loss is the global-batch loss and every
    rank's gradient slices equal the matching rows of the single-process gradients (finetune_repconc.py:296-303)."""
    from repconc_amd.train.stage1 import Stage1Config, global_loss_and_local_grads
    data = _loss_case()
    loss, gq, gp, gn = global_loss_and_local_grads(data["q"], data["pos"], data["neg"], data["qids"], data["pos_ids"],
                                                   data["neg_ids"], data["qrels"],
                                                   Stage1Config(dynamic_topk_hard_negative=5), "METRIC_IP", 8)
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_loss_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        parts = [ret[r] for r in range(world)]
    for r in range(world):
        assert abs(parts[r][0] - float(loss)) < 1e-6
    np.testing.assert_allclose(np.concatenate([p[1] for p in parts]), gq.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.concatenate([p[2] for p in parts]), gp.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.concatenate([p[3] for p in parts]), gn.numpy(), rtol=1e-6, atol=1e-7)


def test_merge_topk_order_and_ties_on_cpu():
    """sharded_search.merge_topk is pure tensor bookkeeping: merged lists are (score desc, id asc), empty slots (-1) sort
    last, and the result equals a brute-force sort of the union."""
    from repconc_amd.sharded_search import merge_topk
    g = torch.Generator().manual_seed(3)
    G, nq, k = 3, 5, 7
    scores = torch.randint(0, 6, (G, nq, k), generator=g).float()            # few distinct values: many ties
    ids = torch.stack([torch.stack([torch.randperm(1000, generator=g)[:k] + 1000 * r for _ in range(nq)]) for r in range(G)])
    scores, order = torch.sort(scores, dim=2, descending=True)
    ids = torch.gather(ids, 2, order)
    scores[2, :, 5:] = float("-inf")
    ids[2, :, 5:] = -1                                                        # a short shard
    ms, mi = merge_topk(scores, ids, k)
    for qi in range(nq):
        items = [(float(scores[r, qi, j]), int(ids[r, qi, j])) for r in range(G) for j in range(k) if int(ids[r, qi, j]) >= 0]
        items.sort(key=lambda t: (-t[0], t[1]))
        assert [int(v) for v in mi[qi]] == [t[1] for t in items[:k]]
        assert [float(v) for v in ms[qi]] == [t[0] for t in items[:k]]


def _rank_sum_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from repconc_amd.train.run_warmup import broadcast_from_rank0_, rank_ordered_sum_
    g = torch.Generator().manual_seed(100 + rank)
    t = torch.randn(64, 64, generator=g, dtype=torch.float64) * (10.0 ** (3 * rank))      # wildly different magnitudes
    init = torch.randn(5, 7, generator=g)
    rank_ordered_sum_(t)
    broadcast_from_rank0_(init)
    ret[rank] = (t.numpy().copy(), init.numpy().copy())
    dist.destroy_process_group()


def test_gloo_rank_ordered_sum_and_rank0_broadcast_are_rank_identical():
    """Multi-rank warm-up (train/run_warmup.py): the Procrustes matrix and the MSE are summed over the ranks in rank
    order and the initial centroids come from rank 0, so every rank holds the same rotation and centroids bit for bit."""
    world = 3
    ret = mp.Manager().dict()
    mp.spawn(_rank_sum_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    parts = [torch.randn(64, 64, generator=torch.Generator().manual_seed(100 + r), dtype=torch.float64) * (10.0 ** (3 * r))
             for r in range(world)]
    want = (parts[0] + parts[1]) + parts[2]
    for r in range(world):
        assert np.array_equal(ret[r][0], want.numpy())          # the same bits on every rank, rank order
        assert np.array_equal(ret[r][1], ret[0][1])


class _BruteIndex:
    """Duck-typed index for the host-side test: exact inner-product top-k over a dense matrix, (score desc, id asc);
    accepts the extra `nprobe` argument an IVF index takes."""

    def __init__(self, base):
        self.base = base

    def search(self, q, k, nprobe=None):
        s = q @ self.base.T
        order = torch.argsort(s, dim=1, descending=True, stable=True)[:, :k]
        return torch.gather(s, 1, order).contiguous(), order.contiguous()


def _replica_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from repconc_amd.sharded_search import replicated_search
    g = torch.Generator().manual_seed(11)                      # the same index and queries on every rank
    index = _BruteIndex(torch.randn(300, 16, generator=g))
    out = {}
    for nq in (7, 2, 0):                                       # ragged split, fewer queries than ranks, none
        q = torch.randn(nq, 16, generator=g)
        s, i = replicated_search(index, q, 5, 8)
        out[nq] = (s.numpy().copy(), i.numpy().copy())
    ret[rank] = out
    dist.destroy_process_group()


def test_gloo_replicated_search_splits_queries_and_gathers_in_order():
    """sharded_search.replicated_search (replica mode of evaluate_repconc.py:131-134 with one process per GPU; also the
    multi-GPU form of the IVF search): every rank ends with the single-index result, in query order."""
    world = 3
    ret = mp.Manager().dict()
    mp.spawn(_replica_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    g = torch.Generator().manual_seed(11)
    index = _BruteIndex(torch.randn(300, 16, generator=g))
    for nq in (7, 2, 0):
        q = torch.randn(nq, 16, generator=g)
        ws, wi = index.search(q, 5)
        for r in range(world):
            s, i = ret[r][nq]
            assert s.shape == (nq, 5) and np.array_equal(i, wi.numpy())
            np.testing.assert_allclose(s, ws.numpy(), rtol=1e-6, atol=1e-6)      # a one-row slice takes the GEMV path: last-ulp
