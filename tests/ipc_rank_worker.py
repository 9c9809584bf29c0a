"""One rank of the multi-process GPU tests (tests/test_ipc_ranks.py launches `world` of these with tests/mp_util.py).

    python tests/ipc_rank_worker.py solve <case> <cut_0> ... <cut_world>     constrained codes of rows [cut_r, cut_r+1)
    python tests/ipc_rank_worker.py recipe8                                   the 49 152-row headline batch over the ranks
    python tests/ipc_rank_worker.py allgather                                 rc_comm_allgather, several sizes
    python tests/ipc_rank_worker.py timeout                                   a missing peer is reported, not waited for
    python tests/ipc_rank_worker.py timeout_solve                             ... inside a whole solve (fused exchange)
    python tests/ipc_rank_worker.py warmup                                    corpus-sharded OPQ + PQ training
    python tests/ipc_rank_worker.py search                                    row-sharded + replicated search gathers
    python tests/ipc_rank_worker.py encode                                    corpus encoding, rows split over the ranks

RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the environment; torch.distributed (gloo) only carries the
set-up handshake, the exchanges themselves run on the IPC transport of librepconc_hip.so.  All ranks share cuda:0 unless
the box has a GPU for each.  Results go to $RC_TEST_OUTDIR/<name>_rank<r>.npy; exit code 0 = this rank's own checks passed.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

EPS, ITERS = 0.003, 100


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = os.environ["RC_TEST_OUTDIR"]
    ndev = torch.cuda.device_count()
    local = rank if ndev >= world else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("gloo", init_method=f"tcp://{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}", rank=rank,
                            world_size=world)
    from repconc_amd import _lib, ops
    what = sys.argv[1]
    rc = 0
    if what == "timeout":
        # rank 0 runs one exchange with a 300 ms limit; the other ranks map the buffers and leave without pushing
        os.environ["RC_IPC_TIMEOUT_MS"] = "300"
        ops.comm_init(transport="ipc")
        if rank == 0:
            flags = torch.zeros(1, dtype=torch.int32, device=dev)
            got = ops.comm_allgather(torch.ones(1024, device=dev), flags)
            torch.cuda.synchronize()
            lib, h = _lib.load(), _lib.handle(local)
            ok = int(flags.item()) & _lib.RC_FLAG_COMM and lib.rc_comm_status(h) & _lib.RC_FLAG_COMM
            # the transport is BROKEN from here on: later exchanges leave at once (no second time-out, no counter re-armed
            # for late arrivals to corrupt) and carry the flag; the torch-facing helpers raise at their check
            import time
            t0 = time.perf_counter()
            f2 = torch.zeros(1, dtype=torch.int32, device=dev)
            for _ in range(5):
                ops.comm_allgather(torch.ones(1024, device=dev), f2)
            torch.cuda.synchronize()
            fast = time.perf_counter() - t0 < 0.25                      # five exchanges, far below ONE 300 ms time-out
            ok = ok and fast and int(f2.item()) & _lib.RC_FLAG_COMM
            ops.all_gather(torch.ones(8, device=dev))
            try:
                ops.comm_check()
                ok = False
            except _lib.RepconcHipError:
                pass
            rc = 0 if ok else 1
            del got
        dist.barrier()
        ops.comm_destroy()
        dist.destroy_process_group()
        return rc
    if what == "timeout_solve":
        # rank 0 runs a whole solve with a 300 ms limit while its peer never calls: every wait of the fused exchange (flag-wait
        # kernels, or the sweeps' prologues with RC_IPC_INWAIT=1) gives up ONCE, the rest of the solve neither waits nor pushes
        os.environ["RC_IPC_TIMEOUT_MS"] = "300"
        ops.comm_init(transport="ipc")
        if rank == 0:
            import time
            xl = torch.randn(512, 768, device=dev)
            Ct = torch.randn(48, 256, 16, device=dev)
            lib, h = _lib.load(), _lib.handle(local)
            t0 = time.perf_counter()
            _, flags = ops.assign_sinkhorn_dist(xl, Ct, EPS, ITERS, torch.uint8)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ok = bool(int(flags.item()) & _lib.RC_FLAG_COMM) and bool(lib.rc_comm_status(h) & _lib.RC_FLAG_COMM) and dt < 5.0
            if not ok:
                print(f"rank 0: flags {int(flags.item())}, status {lib.rc_comm_status(h)}, {dt:.2f} s")
            rc = 0 if ok else 1
        dist.barrier()
        ops.comm_destroy()
        dist.destroy_process_group()
        return rc
    ops.comm_init(transport=os.environ.get("RC_COMM", "ipc"))
    lib, h = _lib.load(), _lib.handle(local)
    assert lib.rc_comm_world(h) == world and lib.rc_comm_kind(h) == (2 if os.environ.get("RC_COMM", "ipc") == "ipc" else 1)

    if what == "solve":
        from conftest import load_case
        name = sys.argv[2]
        cuts = [int(v) for v in sys.argv[3:3 + world + 1]]
        g, x, C = load_case(name)
        xl = torch.from_numpy(x[cuts[rank]:cuts[rank + 1]]).to(dev)
        Ct = torch.from_numpy(C).to(dev)
        want = g["codes_constrained"][cuts[rank]:cuts[rank + 1]]
        # Every form of the exchange, each eager / captured / replayed (/ eager again: the parities keep alternating); all
        # ranks switch between the forms at the same solve.  The small grids of these fixtures fit the one GPU together, so
        # the in-prologue wait (the default when every rank has a GPU of its own) can run between processes that share one;
        # a wait that could not be satisfied ends in RC_FLAG_COMM after the time-out, never in a hang.
        os.environ.setdefault("RC_IPC_TIMEOUT_MS", "30000")
        forms = (("one chain, exchange fused into the sweep, flag-wait kernels (default on a shared GPU)", {}, ("0", "1", "1", "0")),
                 ("one chain, fused, wait inside the next sweep's prologue", {"RC_IPC_INWAIT": "1"}, ("0", "1", "1")),
                 ("two chains, fused", {"RC_DIST_SPLIT": "1"}, ("0", "1", "1")),
                 ("two chains, fused, wait in the prologue", {"RC_DIST_SPLIT": "1", "RC_IPC_INWAIT": "1"}, ("1", "1")),
                 ("two chains, push + wait kernels (rounds 3-4)", {"RC_IPC_XSWEEP": "0"}, ("0", "1", "1")),
                 ("one chain, push + wait kernels", {"RC_IPC_XSWEEP": "0", "RC_DIST_SPLIT": "0"}, ("0", "1", "1")))
        for fi, (form, env, graphs) in enumerate(forms):
            os.environ.update(env)
            for i, graph in enumerate(graphs):
                os.environ["RC_GRAPH"] = graph
                codes, flags = ops.assign_sinkhorn_dist(xl, Ct, EPS, ITERS, torch.uint8)
                torch.cuda.synchronize()
                if fi == 0:
                    np.save(os.path.join(out, f"codes{i}_rank{rank}.npy"), codes.cpu().numpy())
                if int(flags.item()) != 0 or not np.array_equal(codes.cpu().numpy(), want):
                    print(f"rank {rank}: [{form}] pass {i}: flags {int(flags.item())}, "
                          f"{int((codes.cpu().numpy() != want).sum())} codes differ from the golden fixture")
                    rc = 1
            for k in env:
                os.environ.pop(k)
        # an odd iteration count flips the exchange parity between solves: the graph cache must key on it
        os.environ["RC_GRAPH"] = "1"
        ref = None
        for i in range(3):
            codes, _ = ops.assign_sinkhorn_dist(xl, Ct, EPS, 7, torch.uint8)
            torch.cuda.synchronize()
            ref = codes.clone() if ref is None else ref
            if not torch.equal(ref, codes):
                print(f"rank {rank}: 7-iteration solve differs between repetitions ({i})")
                rc = 1
        np.save(os.path.join(out, f"codes7_rank{rank}.npy"), ref.cpu().numpy())
    elif what == "recipe8":
        # BASELINE configs[2]: the headline batch of 49 152 rows on `world` ranks (8: 6 144 rows each), default form of the
        # exchange, eager / captured / replayed; expected = what the REFERENCE returned on eight gloo ranks
        # (tests/golden/recipe8_b49152_m48_sample.npz: identical to its one-process codes)
        from conftest import GOLDEN, load_headline
        x, C, con, _ = load_headline("sample")
        r8 = np.load(os.path.join(GOLDEN, "recipe8_b49152_m48_sample.npz"))
        want_all = con ^ r8["codes_xor_one_process"]
        bl = x.shape[0] // world
        xl = torch.from_numpy(np.ascontiguousarray(x[rank * bl:(rank + 1) * bl])).to(dev)
        del x
        Ct = torch.from_numpy(C).to(dev)
        want = want_all[rank * bl:(rank + 1) * bl]
        os.environ.setdefault("RC_IPC_TIMEOUT_MS", "60000")
        for i, graph in enumerate(("0", "1", "1")):
            os.environ["RC_GRAPH"] = graph
            codes, flags = ops.assign_sinkhorn_dist(xl, Ct, EPS, ITERS, torch.uint8)
            torch.cuda.synchronize()
            bad = int((codes.cpu().numpy() != want).sum())
            if int(flags.item()) != 0 or bad:
                print(f"rank {rank}: pass {i}: flags {int(flags.item())}, {bad} codes differ from the reference's 8-rank run")
                rc = 1
        np.save(os.path.join(out, f"codes_rank{rank}.npy"), codes.cpu().numpy())
    elif what == "allgather":
        for n, dtype in ((7, torch.uint8), (48 * 256, torch.float64), (48 * 256 * 17, torch.float64), (0, torch.float32),
                         (1 << 20, torch.int32)):
            for rep in range(3):
                base = torch.arange(n, device=dev, dtype=torch.float64)
                mine = ((base * (rank + 1) + rep) % 251).to(dtype)
                got = ops.comm_allgather(mine)
                torch.cuda.synchronize()
                for r in range(world):
                    want = ((base * (r + 1) + rep) % 251).to(dtype)
                    if got.shape != (world, n) or not torch.equal(got[r], want):
                        print(f"rank {rank}: all-gather of {n} x {dtype} wrong in slot {r} (rep {rep})")
                        rc = 1
        # the torch-facing helper picks the native layer for CUDA tensors once comm_init() has run
        t = torch.full((5, 3), float(rank), device=dev)
        got = ops.all_gather(t)
        if not all(bool((got[r] == r).all()) for r in range(world)):
            rc = 1
    elif what == "warmup":
        from types import SimpleNamespace
        from repconc_amd.train.run_warmup import warmup_from_embeds
        from oracle import synth
        full = synth.clustered_embeddings(77, 6000)
        cuts = [(6000 * r) // world for r in range(world + 1)]
        model = SimpleNamespace(config=SimpleNamespace(MCQ_M=48, MCQ_K=256, similarity_metric="METRIC_IP"),
                                rotation=torch.eye(768, device=dev),
                                centroids=torch.nn.Parameter(torch.zeros(48, 256, 16, device=dev)))
        model, index = warmup_from_embeds(full[cuts[rank]:cuts[rank + 1]], model, opq_iters=3, pq_iters=4)
        torch.cuda.synchronize()
        np.save(os.path.join(out, f"rotation_rank{rank}.npy"), model.rotation.cpu().numpy())
        np.save(os.path.join(out, f"centroids_rank{rank}.npy"), model.centroids.detach().cpu().numpy())
        np.save(os.path.join(out, f"shard_rank{rank}.npy"), np.array([index.index.id_offset, index.index.ntotal]))
    elif what == "search":
        from repconc_amd.index import PQIndex
        from repconc_amd.sharded_search import replicated_search, sharded_search
        rng = np.random.default_rng(99)
        N, M, nq, k = 40000, 48, 37, 50
        C = rng.standard_normal((M, 256, 16)).astype(np.float32)
        codes = rng.integers(0, 256, (N, M), dtype=np.uint8)
        q = torch.from_numpy(rng.standard_normal((nq, 768)).astype(np.float32)).to(dev)
        whole = PQIndex(768, M, device=dev)
        whole.set_centroids(torch.from_numpy(C))
        whole.add_codes(torch.from_numpy(codes))
        ws, wi = whole.search(q, k)
        cuts = [(N * r) // world for r in range(world + 1)]
        part = PQIndex(768, M, device=dev)
        part.set_centroids(torch.from_numpy(C))
        part.add_codes(torch.from_numpy(codes[cuts[rank]:cuts[rank + 1]]))
        part.id_offset = cuts[rank]
        s1, i1 = sharded_search(part, q, k)
        s2, i2 = replicated_search(whole, q, k)
        torch.cuda.synchronize()
        for nm, (s, i) in (("sharded", (s1, i1)), ("replicated", (s2, i2))):
            if not (torch.equal(s, ws) and torch.equal(i, wi)):
                print(f"rank {rank}: {nm} search differs from the whole-index search")
                rc = 1
    elif what == "encode":
        # evaluate_repconc.py:51-75 + the Trainer's prediction gather: every rank encodes a contiguous share of the corpus
        # to nearest codes, the (padded) shares are gathered; every rank ends with the codes of the WHOLE corpus.
        from types import SimpleNamespace
        from repconc_amd.models.repconc import RepCONC
        from repconc_amd.models.repconc.evaluate_repconc import RepCONCEvaluater
        from oracle import pq_oracle, synth
        n = 1003                                                   # ragged shares
        table = synth.clustered_embeddings(31, n)
        C = synth.sample_centroids(32, table, 48)

        class Enc(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.register_buffer("table", torch.from_numpy(table))
                self.config = SimpleNamespace(hidden_size=768)

            def forward(self, input_ids, attention_mask):
                return self.table[input_ids[:, 0]]

        cfg = SimpleNamespace(MCQ_M=48, MCQ_K=256, hidden_size=768, similarity_metric="METRIC_IP")
        model = RepCONC(cfg, Enc(), False, None, None).to(dev)
        with torch.no_grad():
            model.centroids.copy_(torch.from_numpy(C))
        rows = [{"input_ids": [i, 0], "attention_mask": [1, 1]} for i in range(n)]
        collate = lambda items: {k: torch.tensor([it[k] for it in items]) for k in items[0]}
        args = SimpleNamespace(per_device_eval_batch_size=100, fp16=False, bf16=False)
        for fmt in ("code", "continuous_embedding"):
            got = RepCONCEvaluater(fmt, model=model, args=args, data_collator=collate).predict(rows).predictions
            want = pq_oracle.quantize(table, C, False).astype(np.uint8) if fmt == "code" else table
            if got.shape != want.shape or not (np.array_equal(got, want) if fmt == "code" else np.allclose(got, want, rtol=1e-5, atol=1e-6)):
                print(f"rank {rank}: gathered {fmt} predictions differ from the oracle's")
                rc = 1
    else:
        raise SystemExit(f"unknown scenario {what}")
    ops.comm_destroy()
    dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
