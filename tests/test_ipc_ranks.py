"""The N-rank product path with N PROCESSES on the GPU box (one GPU is enough): rc_pq_assign_sinkhorn_dist and the other
multi-rank gathers on the IPC transport of csrc/comm.hip (rc_comm_ipc_export / _connect / rc_comm_allgather).

Reference behaviour: the dist.is_initialized() branch of RepCONC.quantize (models/repconc/modeling_repconc.py:78-80,
149-157) — every rank's codes equal the codes of the unsharded batch, here the reference-generated golden fixtures.
"""
import numpy as np
import pytest

from conftest import load_case
from mp_util import rank_logs, run_ranks

pytestmark = pytest.mark.gpu


def _cuts(name, world):
    g, _, _ = load_case(name)
    B = int(g["B"])
    if name == "m48_b1000_ragged":                       # two ranks hold 500 rows each, the others are EMPTY ranks
        return B, [0, 500] + [1000] * (world - 1) if world > 1 else [0, B]
    return B, [(B * r) // world for r in range(world + 1)]


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("name", ["m48_b1024_sample", "m8_b2048_sample", "m48_b1000_ragged", "m96_b512_blend"])
def test_native_solve_on_ipc_ranks_equals_golden(name, world, tmp_path):
    """Native C loop with `world` processes, every form of the exchange: ONE launch per iteration (the sweep's reducer pushes
    the row sums, flag-wait kernels or the next sweep's prologue wait; one chain — the default — and two), and the push + wait
    kernels of rounds 3-4 (two chains, one chain) — eager, graph capture, graph replay(, eager again): every pass of every
    rank equals the golden codes of its rows; empty ranks take part in every exchange; a 7-iteration solve (odd: flips the
    exchange parity) is repeatable."""
    B, cuts = _cuts(name, world)
    codes = run_ranks(world, ["solve", name] + cuts, str(tmp_path), timeout=420)
    assert codes == [0] * world, rank_logs(str(tmp_path), world)
    g, _, _ = load_case(name)
    for i in range(4):
        got = np.concatenate([np.load(tmp_path / f"codes{i}_rank{r}.npy") for r in range(world)], 0)
        assert np.array_equal(got, g["codes_constrained"]), (name, world, i)


def test_the_eight_rank_recipe_on_ipc_ranks_equals_the_reference(tmp_path):
    """BASELINE configs[2] as far as one GPU can go: the 49 152 x 768 headline batch, M = 48, on EIGHT rank processes of
    6 144 rows each (the 8-GPU recipe's per-rank shape and exchange count), native C loop, default form of the exchange,
    eager / captured / replayed.  Expected values: what /root/reference returned when eight gloo ranks ran its distributed
    branch on this batch (oracle/gen_golden.py --recipe8; its codes equal its own one-process codes in all 2 359 296 places)."""
    import os
    from conftest import GOLDEN, load_headline
    r8 = np.load(os.path.join(GOLDEN, "recipe8_b49152_m48_sample.npz"))
    assert int(r8["world"]) == 8 and not r8["codes_xor_one_process"].any()
    world = 8
    assert run_ranks(world, ["recipe8"], str(tmp_path), timeout=900) == [0] * world, rank_logs(str(tmp_path), world)
    _, _, con, _ = load_headline("sample")
    got = np.concatenate([np.load(tmp_path / f"codes_rank{r}.npy") for r in range(world)], 0)
    assert np.array_equal(got, con)


def test_seven_iteration_solve_on_ipc_ranks_equals_single_process(tmp_path):
    import torch
    from repconc_amd import ops
    name, world = "m48_b1024_sample", 2
    B, cuts = _cuts(name, world)
    assert run_ranks(world, ["solve", name] + cuts, str(tmp_path), timeout=420) == [0] * world, rank_logs(str(tmp_path), world)
    _, x, C = load_case(name)
    want, _ = ops.assign_sinkhorn(torch.from_numpy(x).cuda(), torch.from_numpy(C).cuda(), 0.003, 7, torch.uint8)
    got = np.concatenate([np.load(tmp_path / f"codes7_rank{r}.npy") for r in range(world)], 0)
    assert np.array_equal(got, want.cpu().numpy())


@pytest.mark.parametrize("world", [2, 3])
def test_ipc_allgather_sizes(world, tmp_path):
    """rc_comm_allgather: 7 bytes, one exchange slot, several slots (chunked), nothing, 4 MiB — three times each."""
    assert run_ranks(world, ["allgather"], str(tmp_path), timeout=240) == [0] * world, rank_logs(str(tmp_path), world)


def test_ipc_wait_gives_up_on_a_missing_peer(tmp_path):
    """A peer that never pushes: the wait kernel leaves after RC_IPC_TIMEOUT_MS and raises RC_FLAG_COMM (flags word and
    rc_comm_status) instead of hanging the queue."""
    assert run_ranks(2, ["timeout"], str(tmp_path), timeout=240) == [0, 0], rank_logs(str(tmp_path), 2)


@pytest.mark.parametrize("inwait", ["0", "1"])
def test_fused_exchange_gives_up_on_a_missing_peer(inwait, tmp_path, monkeypatch):
    """A whole solve whose peer never calls: the first wait of the fused exchange (flag-wait kernel / the sweep's prologue)
    leaves after RC_IPC_TIMEOUT_MS, the transport is broken from then on (no later wait, no later push), the solve returns
    within seconds with RC_FLAG_COMM in its flags."""
    monkeypatch.setenv("RC_IPC_INWAIT", inwait)
    assert run_ranks(2, ["timeout_solve"], str(tmp_path), timeout=240) == [0, 0], rank_logs(str(tmp_path), 2)


def test_corpus_sharded_warmup_on_ipc_ranks_is_rank_identical(tmp_path):
    """BASELINE configs[2] ("corpus sharded ... centroid all-gather"): OPQ + PQ training with the corpus rows split over
    two processes — per-shard Lloyd statistics and Procrustes matrices gathered through rc_comm_allgather, summed in rank
    order: both ranks end with bit-identical rotations and centroids; the shards are contiguous."""
    world = 2
    assert run_ranks(world, ["warmup"], str(tmp_path), timeout=420) == [0] * world, rank_logs(str(tmp_path), world)
    r0, r1 = np.load(tmp_path / "rotation_rank0.npy"), np.load(tmp_path / "rotation_rank1.npy")
    c0, c1 = np.load(tmp_path / "centroids_rank0.npy"), np.load(tmp_path / "centroids_rank1.npy")
    assert np.array_equal(r0.view(np.uint32), r1.view(np.uint32)) and np.array_equal(c0.view(np.uint32), c1.view(np.uint32))
    assert np.abs(r0 @ r0.T - np.eye(768)).max() < 1e-4 and np.isfinite(c0).all() and np.abs(c0).max() > 0
    s0, s1 = np.load(tmp_path / "shard_rank0.npy"), np.load(tmp_path / "shard_rank1.npy")
    assert list(s0) == [0, 3000] and list(s1) == [3000, 3000]


def test_sharded_and_replicated_search_gathers_on_ipc_ranks(tmp_path):
    """evaluate_repconc.py:121-135 in the one-process-per-rank model: row-sharded search (local top-k, gather, merge) and
    replica search (query slices, gather) with the result gathers on the IPC layer equal the whole-index search."""
    world = 3
    assert run_ranks(world, ["search"], str(tmp_path), timeout=300) == [0] * world, rank_logs(str(tmp_path), world)


def test_corpus_encoding_split_over_ipc_ranks(tmp_path):
    """Corpus encoding with the rows split over three processes (ragged shares, a partial last batch): the shares' codes /
    embeddings are gathered on the IPC layer; every rank holds the oracle's nearest codes of the whole corpus."""
    world = 3
    assert run_ranks(world, ["encode"], str(tmp_path), timeout=300) == [0] * world, rank_logs(str(tmp_path), world)


def test_bench_gpus2_typed_as_is_on_a_shared_gpu():
    """`python bench.py --gpus 2` without a launcher (how the driver types it): bench.py spawns its two ranks; with
    RC_BENCH_SHARE_GPU=1 they share cuda:0, the handshake runs over gloo and the exchange over the IPC transport.  ONE JSON
    line, n_gpus 2, native C loop, sharded codes == unsharded codes."""
    import json
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    if torch.cuda.device_count() < 2:
        env["RC_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu",
                        "--no-adc", "--batch", "8192"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    chk = d["multi_gpu_check"]
    assert chk["sharded_equals_unsharded"] is True and chk["transport"] in ("ipc", "ipc-kernels", "rccl") and chk["native_equals_staged"] is True
    assert d["exchange"]["us_per_allgather"] > 0


def test_first_contact_tool_walks_every_transport_on_the_shared_gpu():
    """tools/first_contact.py — the command for the first run on a multi-GPU node — on this box: every IPC allocation kind
    sets up, moves intact data and is timed (with >= 2 GPUs it runs one rank per GPU and RCCL too); a broken candidate is
    named, not hung on."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "first_contact.py"), "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    table = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert table["ipc/finegrained"]["ok"] and table["ipc/finegrained"]["us_per_allgather"] > 0
    assert all(v["ok"] or v["reason"] for v in table.values())
