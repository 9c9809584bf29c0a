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
