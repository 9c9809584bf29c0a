#!/usr/bin/env python3
"""Headline benchmark of the RepCONC PQ hot path on MI355X (BASELINE.json metric:
"constrained-cluster assignments/sec + ADC queries/sec, 8.8M x 768d M=48 K=256").

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one constrained (Sinkhorn, eps=0.003, 100 iterations) code assignment of one training
batch of B_global = 49152 document embeddings (4096 queries x (1 positive + 11 negatives),
examples/sentence-bert/repconc/7_run_conc_train.sh:18-22) at D=768, M=48, K=256 — SURVEY.md §8d
workload B, BASELINE.json configs[1].  180 such batches = the 8.84M-passage corpus.  The batch is
resident in HBM (fp32, synthetic N(0,1)) before the timed region.  With N ranks, rank r owns rows
[r*B/N, (r+1)*B/N) and the row sums are all-gathered every iteration (strong scaling of a step).

`value` = constrained assignments (document vectors) per second, whole job.  The ADC leg (second
half of the metric) is reported in the `adc` object of the same JSON line: PQ inner-product top-1000
search of 1200-query batches over a resident 8,841,823 x 48-byte index; with N ranks the index is
replicated and the queries are split (the reference's co.shard=False, evaluate_repconc.py:131-134).

`roofline` is for the dominant kernel, the Sinkhorn sweep (sk_pass_kernel<false>): algorithmic
bytes per launch = B_local*M*K*4 (one fp32 read of the K centred distances of every (vector,
sub-quantiser)) divided by its average launch duration, measured with HIP events
recorded around every launch inside the timed region (rc_profile_*, on the launch stream).
`cpu_baseline` times the oracle's C port of the reference algorithm on the host cores (rank 0,
N=1 only) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D, M, K = 768, 48, 256
B_GLOBAL = 49152
EPS, ITERS = 0.003, 100
N_CORPUS = 8841823
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-adc", action="store_true", help="skip the ADC leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-opq", action="store_true",
                    help="skip the OPQ / PQ training leg (profiling runs: its many small launches of the assignment "
                         "kernels would dilute the per-kernel means of the index-build leg)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the live rocprofv3 --pmc passes of roofline.traffic (profiling runs: bench.py is then already under "
                         "rocprofv3); the figure of profiles/pmc_summary.json is reported instead")
    ap.add_argument("--no-per-rank", action="store_true",
                    help="skip the per_rank_6144 leg (profiling runs: its launches of the sweep kernel would mix into "
                         "the per-kernel averages of the timed 49152-row configuration)")
    ap.add_argument("--adc-batches", type=int, default=6,
                    help="1200-query batches in the timed ADC region (6 = the 6980 dev queries of the reference's evaluation)")
    ap.add_argument("--adc-k", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=B_GLOBAL, help=argparse.SUPPRESS)
    ap.add_argument("--force-dist", action="store_true",
                    help="development: run the N>1 code path (RCCL collectives, staged sweeps) with one rank")
    ap.add_argument("--dist-driver", choices=("auto", "staged"), default="auto",
                    help="N>1: auto = native RCCL loop if it passes the self-check, staged = torch.distributed loop")
    return ap.parse_args()


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed PMC summary (profiles/pmc_summary.json), else None."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        return json.load(open(p))[kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def measure_sweep_traffic(B):
    """HBM bytes per launch of the dominant kernel MEASURED BY THIS RUN: two `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE — the
    two do not fit one pass; counters only, never combined with a trace domain other than the kernel trace, as MI355X_MICROARCH.md's
    HBM section prescribes) over tools/sweep_traffic_child.py, which solves one batch of this run's shape twice with eager launches.
    FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (the guide's gfx950 correction).  None if rocprofv3 is missing or a pass
    fails — the line then keeps the figure of profiles/pmc_summary.json and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    child = os.path.join(ROOT, "tools", "sweep_traffic_child.py")
    means, launches = {}, None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        work = tempfile.mkdtemp(prefix="rc_pmc_")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", work, "-o", "p", "--",
                                sys.executable, child, str(B)], cwd="/tmp", env=env, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=240)
            if r.returncode != 0:
                return None
            vals = []
            for f in glob.glob(os.path.join(work, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and "sk_sweep2_kernel<2, true" in row.get("Kernel_Name", ""):
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None
            means[counter], launches = sum(vals) / len(vals), len(vals)
        except Exception:
            return None
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return {"bytes_per_launch": int(2 * means["FETCH_SIZE"] * 1024 + means["WRITE_SIZE"] * 1024), "launches": launches,
            "fetch_size_kib": round(means["FETCH_SIZE"], 1), "write_size_kib": round(means["WRITE_SIZE"], 1)}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` typed as is (no launcher, WORLD_SIZE unset): start N ranks of this script, one per GPU,
    over 127.0.0.1; rank 0 prints the JSON line.  Fewer than N GPUs visible: one {"skipped": ...} line, exit 0
    (RC_BENCH_SHARE_GPU=1 puts every rank on cuda:0 instead — a walk through the N-rank code, flagged in the line).
    A rank that dies takes the others down (exact PIDs) and its exit code becomes this process's."""
    import socket
    import subprocess
    import torch
    n_vis = torch.cuda.device_count()
    share = os.environ.get("RC_BENCH_SHARE_GPU", "0") == "1"
    if n_vis < args.gpus and not share:
        print(json.dumps({"skipped": f"--gpus {args.gpus} requested but {n_vis} GPU(s) visible (RC_BENCH_SHARE_GPU=1 runs "
                                     "all ranks on one GPU as a code walk)", "metric": "constrained_cluster_assignments_per_sec",
                          "value": None, "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup}), flush=True)
        return 0
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = set(range(args.gpus))
        while live:
            for r in list(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 1
            if rc:
                break
            time.sleep(0.2)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
                pr.wait()
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    # Only the JSON line may appear on stdout: libraries loaded below (RCCL prints a version banner) write to
    # fd 1 from C, so fd 1 is pointed at stderr for the duration and the result goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from repconc_amd import _lib, ops
    from repconc_amd.sharded import SingleComm, TorchDistComm, assign_sinkhorn_sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # RC_BENCH_SHARE_GPU=1 (development only, flagged in the line): every rank on cuda:0 with gloo collectives, to walk
    # the N > 1 branches of this file on a one-GPU box; RCCL refuses two ranks on one device, so the native solve's probe
    # fails there and the staged driver runs.  The numbers of such a run mean nothing.
    share_gpu = os.environ.get("RC_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        comm = TorchDistComm()
    else:
        comm = SingleComm()
    lib, h = _lib.load(), _lib.handle(local_rank)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if not use_dist:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ------------------------------------------------------------------ constrained assignment
    B = args.batch
    assert B % world == 0
    bl = B // world
    n_pool = 3                                              # distinct batches cycled through
    rng = np.random.default_rng(20220)
    pool = []
    for i in range(n_pool):
        xb = rng.standard_normal((B, D), dtype=np.float32)
        if i == 0:
            cent = np.ascontiguousarray(
                xb[np.random.default_rng(20221).permutation(B)[:K]].reshape(K, M, D // M).transpose(1, 0, 2))
        pool.append(torch.from_numpy(xb[rank * bl:(rank + 1) * bl]).to(dev))
    C = torch.from_numpy(cent).to(dev)

    def step(i):
        x = pool[i % n_pool]
        if not use_dist:
            return ops.assign_sinkhorn(x, C, EPS, ITERS, torch.uint8)
        return assign_sinkhorn_sharded(x, C, EPS, ITERS, comm, dtype=torch.uint8)

    # ------------------------------------------------------------------ multi-rank self-check (untimed)
    # On a small global batch: the Python-staged torch.distributed solve is the yardstick; the native C loop
    # (csrc/comm.hip) is tried with each exchange transport in turn — ipc (peer stores into IPC-mapped buffers; works on
    # a shared GPU too), then rccl — first in a child process with a timeout (a hang or crash in a transport this node has
    # never run stays there), then in-process against the staged codes.  The first transport every rank passes is used;
    # if none, every rank times the staged driver and the line says why.  (2) the gathered sharded codes against the
    # unsharded single-GPU solve of the same batch on rank 0.
    dist_check = None
    exchange = None
    if use_dist:
        gb = 1024 * world
        xs_full = np.random.default_rng(20230).standard_normal((gb, D), dtype=np.float32)
        xs_loc = torch.from_numpy(xs_full[rank * 1024:(rank + 1) * 1024]).to(dev)
        os.environ["RC_DIST_NATIVE"] = "0"
        c_staged, _ = assign_sinkhorn_sharded(xs_loc, C, EPS, ITERS, comm, dtype=torch.uint8)

        def all_ranks(ok):
            t = torch.tensor([int(ok)], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        chosen, notes, measured = None, {}, {}
        if args.dist_driver == "staged":
            notes["staged"] = "--dist-driver staged"
            candidates = []
        elif os.environ.get("RC_COMM", "auto").lower() in ("ipc", "rccl"):
            candidates = [os.environ["RC_COMM"].lower()]
        else:
            # "ipc" = the exchange fused into the sweep kernels (round 5: peer stores from the sweep's reducer, the wait in the next
            # sweep's prologue); "ipc-kernels" = the same transport with the push + wait kernels of rounds 3-4 (RC_IPC_XSWEEP=0),
            # tried only when the fused form does not pass on this node; RCCL refuses two ranks on one device
            candidates = ["ipc", "ipc-kernels"] if share_gpu else ["ipc", "ipc-kernels", "rccl"]
        os.environ["RC_COMM_STRICT"] = "1"                  # "ipc" means ipc here: no silent fall-back inside comm_init
        import subprocess
        XSWEEP_ENV = os.environ.get("RC_IPC_XSWEEP")

        def transport_env(label):
            """(RC_COMM value, RC_IPC_XSWEEP value or None) of a candidate"""
            return ("ipc", "0") if label == "ipc-kernels" else (label, XSWEEP_ENV)
        for ti, label in enumerate(candidates):
            if label == "ipc-kernels" and "ipc" in measured:
                continue                                     # the fused form works here: nothing to fall back from
            transport, xs = transport_env(label)
            if xs is None:
                os.environ.pop("RC_IPC_XSWEEP", None)
            else:
                os.environ["RC_IPC_XSWEEP"] = xs
            # the probe's exchange waits give up after 8 s (RC_FLAG_COMM), the probe itself after 150 s: a transport that does
            # not work on this node costs minutes at most, never the run
            env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29533")) + 17 + ti), RC_COMM=transport,
                       RC_IPC_TIMEOUT_MS=os.environ.get("RC_IPC_TIMEOUT_MS", "8000"))
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)     # the child makes its own TCP store on the new port
            child = subprocess.Popen([sys.executable, "-m", "repconc_amd.dist_probe"], cwd=ROOT, env=env,
                                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            try:
                prc = child.wait(timeout=150)
            except subprocess.TimeoutExpired:
                child.kill()
                child.wait()
                prc = -9
            if not all_ranks(prc == 0):
                notes[label] = f"out-of-process probe failed (exit {prc} on this rank)"
                continue
            os.environ["RC_DIST_NATIVE"], os.environ["RC_COMM"] = "1", transport
            ok, why = True, ""
            try:
                c_native, _ = assign_sinkhorn_sharded(xs_loc, C, EPS, ITERS, comm, dtype=torch.uint8)
                torch.cuda.synchronize()
                if not torch.equal(c_native, c_staged):
                    ok, why = False, "codes differ from the staged driver"
            except Exception as e:                           # transport set-up failure: reported, not hidden
                ok, why = False, f"{type(e).__name__}: {e}"
            if not all_ranks(ok):
                notes[label] = why or "failed on another rank"
                try:
                    ops.comm_destroy()
                except Exception:
                    pass
                continue
            # this transport works: what it costs.  (a) one exchange step alone: all-gather of a chain's [M/2, K] fp64 row
            # sums, 200 back to back; (b) the critical path it sits on: two untimed-warm steps of THIS rank's share of
            # the batch through the native loop (sweeps of one chain overlap the exchange of the other)
            rows = torch.zeros((M // 2, K), dtype=torch.float64, device=dev)
            for _ in range(20):
                ops.comm_allgather(rows)
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                ops.comm_allgather(rows)
            e1.record()
            torch.cuda.synchronize()
            us_ag = max_over_ranks(e0.elapsed_time(e1)) * 1e3 / 200
            # ... as TWO chains of M/2 sub-quantisers (RC_DIST_SPLIT=1: the sweep of one chain overlaps the exchange of the
            # other; the library default on RCCL) and as ONE chain (RC_DIST_SPLIT=0; the library default on the IPC transport,
            # where the exchange is part of the sweep kernel: the reducer of a sub-quantiser pushes its row sums, the next
            # sweep's prologue waits).  Which is faster depends on what an exchange costs on this node against a second launch
            # per iteration; the node decides, the line says which ran.
            by_chains = {}
            for split in ("1", "0"):
                os.environ["RC_DIST_SPLIT"] = split
                for _ in range(2):
                    step(0)
                barrier()
                tt0 = time.perf_counter()
                for i in range(3):
                    step(i)
                barrier()
                by_chains[split] = max_over_ranks(time.perf_counter() - tt0) * 1e3 / 3
            split_best = min(by_chains, key=lambda k: by_chains[k])
            ms_step = by_chains[split_best]
            measured[label] = {"us_per_allgather": round(us_ag, 2), "ms_per_step": round(ms_step, 3),
                                   "us_per_iteration": round(ms_step * 1e3 / ITERS, 2),
                                   "chains": 2 if split_best == "1" else 1,
                                   "ms_per_step_two_chains": round(by_chains["1"], 3),
                                   "ms_per_step_one_chain": round(by_chains["0"], 3)}
            ops.comm_check()
            barrier()
            try:
                ops.comm_destroy()
            except Exception:
                pass
        if measured:                                         # the faster one runs the timed region (same choice on every rank:
            chosen = min(measured, key=lambda t: measured[t]["ms_per_step"])     # ms_per_step is a max over ranks)
            c_transport, c_xs = transport_env(chosen)
            os.environ["RC_DIST_NATIVE"], os.environ["RC_COMM"] = "1", c_transport
            if c_xs is None:
                os.environ.pop("RC_IPC_XSWEEP", None)
            else:
                os.environ["RC_IPC_XSWEEP"] = c_xs
            os.environ["RC_DIST_SPLIT"] = "1" if measured[chosen]["chains"] == 2 else "0"
        native_all = chosen is not None
        os.environ["RC_DIST_NATIVE"] = "1" if native_all else "0"
        gathered = [torch.empty_like(c_staged) for _ in range(world)]
        dist.all_gather(gathered, c_staged)
        unsharded_equal = None
        if rank == 0:
            ref_codes, _ = ops.assign_sinkhorn(torch.from_numpy(xs_full).to(dev), C, EPS, ITERS, torch.uint8)
            unsharded_equal = bool(torch.equal(torch.cat(gathered, 0), ref_codes))
        dist_check = {"global_batch": gb,
                      "driver": (f"native C loop (csrc/comm.hip), {chosen} transport" if native_all else
                                 "python-staged torch.distributed loop"),
                      "transport": chosen, "native_equals_staged": native_all if candidates else None,
                      "transport_notes": notes or None, "sharded_equals_unsharded": unsharded_equal}
        # (3) the recipe's own batch against the REFERENCE: the 49 152-row batch of
        # tests/golden/headline_b49152_m48_sample.npz, `world` equal row blocks through the driver chosen above, compared with
        # what the reference's distributed branch returned on eight gloo ranks (recipe8_b49152_m48_sample.npz — identical to
        # its one-process codes, so the expected codes do not depend on the number of ranks)
        golden = os.path.join(ROOT, "tests", "golden")
        try:
            import zlib
            from oracle import synth                         # input synthesis from seeds (shared with the fixtures), no arithmetic
            hg = np.load(os.path.join(golden, "headline_b49152_m48_sample.npz"))
            r8 = np.load(os.path.join(golden, "recipe8_b49152_m48_sample.npz"))
            Bh = int(hg["B"])
            if Bh % world == 0:
                xh = synth.clustered_embeddings(int(hg["x_seed"]), Bh, n_clusters=int(hg["n_clusters"]))
                Ch = hg["centroids"] if "centroids" in hg.files else synth.sample_centroids(int(hg["c_seed"]), xh, M)
                assert zlib.crc32(np.ascontiguousarray(Ch).tobytes()) == int(hg["centroids_crc"])
                want = hg["codes_constrained"] ^ r8["codes_xor_one_process"]
                blh = Bh // world
                xh_l = torch.from_numpy(np.ascontiguousarray(xh[rank * blh:(rank + 1) * blh])).to(dev)
                del xh
                ch_l, fl_h = assign_sinkhorn_sharded(xh_l, torch.from_numpy(np.ascontiguousarray(Ch)).to(dev), EPS, ITERS, comm,
                                                     dtype=torch.uint8)
                torch.cuda.synchronize()
                bad_h = int((ch_l.cpu().numpy() != want[rank * blh:(rank + 1) * blh]).sum()) + (0 if int(fl_h.item()) == 0 else 1)
                dist_check["recipe_batch_equals_the_references_8_rank_run"] = all_ranks(bad_h == 0)
                dist_check["recipe_batch"] = f"{Bh} x 768, M=48: {blh} rows per rank on {world} ranks, every code compared"
                del xh_l, ch_l
        except FileNotFoundError as e:
            dist_check["recipe_batch_equals_the_references_8_rank_run"] = None
            dist_check["recipe_batch"] = f"fixture missing: {e}"
        # (4) who sits where: one line per rank (device index, PCI bus id), and whether every rank has a device of its own — an
        # IPC transport that passed above between ranks on DISTINCT devices has mapped a peer device's buffer
        # (hipIpcOpenMemHandle across devices) and exchanged through it
        prop = torch.cuda.get_device_properties(dev)
        ident = f"{local_rank}:{getattr(prop, 'pci_bus_id', '?')}:{getattr(prop, 'pci_device_id', '?')}:{getattr(prop, 'uuid', '')}"
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        distinct = len(set(idents)) == world
        dist_check["rank_devices"] = idents
        dist_check["one_device_per_rank"] = distinct
        dist_check["ipc_between_peer_devices"] = (bool(distinct and any(t.startswith("ipc") for t in measured)) if candidates else None)
        # (5) the ceiling of this rank count: this rank's share of the batch solved stand-alone (no exchange at all)
        for _ in range(2):
            ops.assign_sinkhorn(pool[0], C, EPS, ITERS, torch.uint8)
        barrier()
        tt0 = time.perf_counter()
        for i in range(3):
            ops.assign_sinkhorn(pool[i % n_pool], C, EPS, ITERS, torch.uint8)
        barrier()
        alone_ms = max_over_ranks(time.perf_counter() - tt0) * 1e3 / 3
        dist_check["standalone_share"] = {"rows": bl, "ms_per_step": round(alone_ms, 3),
                                          "ceiling_vectors_per_sec": round(B / (alone_ms * 1e-3), 1),
                                          "what": "every rank solves its own rows with NO exchange (a different problem: the "
                                                  "constraint is per rank): the time a sharded step cannot beat"}
        if native_all:
            exchange = {"transport": chosen, "chains": measured[chosen]["chains"],
                        "bytes_per_rank": (M // measured[chosen]["chains"]) * K * 8,
                        "us_per_allgather": measured[chosen]["us_per_allgather"],
                        "per_transport": measured,
                        "what": "per transport that passed the probe and reproduced the staged codes: 200 back-to-back "
                                "rc_comm_allgather calls of one chain's row sums (fused push + wait kernel, copy-out), and three "
                                "steps of this run's per-rank batch through the native loop (ms_per_step, us_per_iteration = one "
                                "Sinkhorn iteration's critical path) as two chains of M/2 sub-quantisers (RC_DIST_SPLIT=1) and as "
                                "one chain (RC_DIST_SPLIT=0; the IPC transport's default: ONE launch per iteration, the sweep's "
                                "reducer pushes the row sums to every peer and the next sweep's prologue - a flag-wait kernel when "
                                "ranks share a device - waits for theirs); the (transport, chains) pair with the shortest step "
                                "runs the timed region"}
            barrier()
        del xs_loc, c_staged, gathered

    for i in range(args.warmup):
        codes, flags = step(i)
    barrier()
    import ctypes
    # The timed region runs the product default: sweeps t >= 2 (and, N > 1, their all-gathers) replayed from the captured
    # hipGraph.  N = 1: each step's run of T - 2 = 98 sweep launches is bracketed by ONE pair of HIP events on the launch
    # stream (rc_profile_enable(h, 2)): avg_launch_ms = bracket time / 98, the ~1.5 us gaps between launches included.
    # N > 1 (two chains on two streams overlap): the per-launch time comes from ONE extra, untimed step with an event
    # pair around every launch (eager loop).
    profile_timed = not use_dist
    if profile_timed:
        lib.rc_profile_enable(h, 2)
    t0 = time.perf_counter()
    for i in range(args.steps):
        codes, flags = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if not profile_timed:
        lib.rc_profile_enable(h, 1)
        step(args.warmup + args.steps)
        barrier()
    lib.rc_profile_enable(h, 0)
    n_l, ms_l = ctypes.c_int(0), ctypes.c_double(0.0)
    lib.rc_profile_collect(h, _lib.PROF_SK_PASS, ctypes.byref(n_l), ctypes.byref(ms_l))
    dt = max_over_ranks(dt)
    assert int(flags.item()) == 0, "Sinkhorn produced non-finite sums"
    value = args.steps * B / dt
    sweep_ms = ms_l.value / max(n_l.value, 1)
    # with N > 1 ranks the sub-quantisers run as two chains: one launch covers M/2 of them
    if not use_dist or os.environ.get("RC_DIST_NATIVE", "1") != "0":
        n_chains = lib.rc_solve_num_chains_on(h, world, M)
    else:                                                   # staged driver: two halves when world > 1 (sharded.py)
        n_chains = 2 if (world > 1 and M >= 2 and os.environ.get("RC_SHARD_SPLIT", "1") != "0") else 1
    alg_bytes = bl * (M // n_chains) * K * 4
    achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9 if n_l.value else 0.0
    roofline = {"kernel": "sk_sweep2_kernel<2, true> (Sinkhorn sweep t >= 1 incl. fused row-potential update and integer "
                          "column exponents; potentials from LDS, 4 blocks per CU, one equal column range per block, wave "
                          "priority rotating with the clock)",
                "measured_in": "timed region: one HIP-event pair around each step's 98 back-to-back sweep launches (t >= 2, "
                               "replayed from the hipGraph), divided by 98 - inter-launch gaps included" if profile_timed else
                               "one extra profiled step after the timed region (the timed region replays the hipGraph)",
                "bound": "hbm", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic("sk_sweep_kernel"),
                "traffic_source": "profiles/pmc_summary.json (rocprofv3 --pmc passes of an earlier profiled run of this "
                                  "command), NOT measured by this run",
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(sweep_ms, 4), "launches_timed": n_l.value,
                "sub_quantisers_per_launch": M // n_chains,
                "measured_read_ceiling": {"GBs": 6850, "with_the_sweeps_valu_load_GBs": [6290, 6490],
                                          "note": "tools/ubench_stream_read.hip (profiles/r04e_ubench_stream_read.txt, not this "
                                                  "run): a kernel that only reads the table with this access pattern reaches "
                                                  "6.85 TB/s = 0.86 of the nominal peak used in `frac`; with 12-16 dependent "
                                                  "fp64 FMAs per entry (the sweep issues ~16 VALU instructions per entry) "
                                                  "6.3-6.5 TB/s"}}

    # balance sanity of the last batch (every centroid gets ~B/K of the global batch)
    hist = ops.code_hist(codes)
    if use_dist:
        dist.all_reduce(hist)
    ideal = B / K
    imb = float((hist.float() / ideal - 1).abs().max().item())

    out = {
        "metric": "constrained_cluster_assignments_per_sec", "value": round(value, 1), "unit": "vectors/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "SURVEY 8d-B / BASELINE configs[1] shape: constrained PQ assignment of "
                               "49152x768 batches (180 = 8.84M corpus), M=48 K=256 eps=0.003 T=100",
                   "global_batch": B, "rows_per_gpu": bl, "D": D, "M": M, "K": K, "sk_iters": ITERS,
                   "parallelism": f"batch-sharded x{world}, all-gather of [M,K] f64 row sums per iteration"
                                  + ((f", {os.environ.get('RC_COMM', 'ipc')} exchange driven from C, "
                                      + ("two chains of M/2 sub-quantisers on two streams" if n_chains == 2 else
                                         "one chain, the exchange inside the sweep kernel")
                                      if os.environ.get("RC_DIST_NATIVE", "1") != "0" else
                                      ", python-staged torch.distributed loop (the native driver did not pass the probe)")
                                     if use_dist else "")},
        "sub_assignments_per_sec": round(value * M, 1),
        "max_code_imbalance": round(imb, 4),
        "roofline": roofline,
    }
    if dist_check is not None:
        out["multi_gpu_check"] = dist_check
    if exchange is not None:
        out["exchange"] = exchange
    if share_gpu:
        out["test_mode"] = ("RC_BENCH_SHARE_GPU=1: all ranks on ONE GPU (gloo handshake, IPC exchange between the processes) - a "
                            "walk through the N > 1 code, not a measurement")

    # ------------------------------------------------------------------ the 8-GPU recipe's per-rank shape on this GPU
    if not use_dist and B == B_GLOBAL and not args.no_per_rank:
        blr = B_GLOBAL // 8
        xr = pool[0][:blr].contiguous()
        for _ in range(2):
            ops.assign_sinkhorn(xr, C, EPS, ITERS, torch.uint8)
        torch.cuda.synchronize()
        lib.rc_profile_enable(h, 2)
        t0 = time.perf_counter()
        nrep = 10
        for _ in range(nrep):
            ops.assign_sinkhorn(xr, C, EPS, ITERS, torch.uint8)
        torch.cuda.synchronize()
        rdt = (time.perf_counter() - t0) / nrep
        lib.rc_profile_enable(h, 0)
        lib.rc_profile_collect(h, _lib.PROF_SK_PASS, ctypes.byref(n_l), ctypes.byref(ms_l))
        rs_ms = ms_l.value / max(n_l.value, 1)
        r_alg = blr * M * K * 4
        out["per_rank_6144"] = {
            "what": "one rank's share of the 8-GPU recipe (6144 x 768, M=48) solved stand-alone on this GPU: no "
                    "collective in it, so 8 x value is the ceiling of the 8-GPU run before any all-gather latency",
            "value": round(blr / rdt, 1), "unit": "vectors/s", "ms_per_step": round(rdt * 1e3, 3),
            "roofline": {"kernel": "sk_sweep2_kernel<2, true>", "bound": "hbm", "achieved": round(r_alg / (rs_ms * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(r_alg / (rs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes_per_launch": r_alg, "avg_launch_ms": round(rs_ms, 4),
                         "launches_timed": n_l.value}}

    # ------------------------------------------------------------------ ADC search leg
    if not args.no_adc:
        del pool
        torch.cuda.empty_cache()
        gen = torch.Generator(device=dev).manual_seed(20222)
        index_codes = torch.randint(0, 256, (N_CORPUS, M), dtype=torch.uint8, device=dev, generator=gen)
        nq_batch = 1200
        q_all = torch.from_numpy(np.random.default_rng(20223).standard_normal((nq_batch * args.adc_batches, D),
                                                                             dtype=np.float32)).to(dev)
        per_rank = nq_batch // world
        k = args.adc_k

        from repconc_amd.index import PQIndex
        index = PQIndex(D, M, device=dev)                      # keeps the permuted code image next to the codes
        index.set_centroids(C)
        index.add_codes(index_codes)
        index_codes = index.codes

        def search(bi):
            q = q_all[bi * nq_batch + rank * per_rank: bi * nq_batch + (rank + 1) * per_rank]
            return index.search_async(q, k)

        search(0)()
        barrier()
        lib.rc_profile_enable(h, 1)
        t0 = time.perf_counter()
        pending = []
        for bi in range(args.adc_batches):                           # as batch_search: every batch enqueued, then read
            pending.append((search(bi), index.last_search))
        adc_repeated = adc_exact = 0
        for fin, pend_ in pending:
            sc, ids = fin()
            adc_repeated += pend_.stats["retried_queries"]
            adc_exact += pend_.stats["exact_queries"]
        barrier()
        adt = max_over_ranks(time.perf_counter() - t0)
        lib.rc_profile_enable(h, 0)
        lib.rc_profile_collect(h, _lib.PROF_ADC_SCAN, ctypes.byref(n_l), ctypes.byref(ms_l))
        qps = args.adc_batches * per_rank * world / adt
        scan_ms = ms_l.value / max(n_l.value, 1)
        adc_alg = per_rank * N_CORPUS * M          # N*M code bytes per query
        adc_ach = adc_alg / (scan_ms * 1e-3) / 1e9 if n_l.value else 0.0
        # LDS gather roof ("then the bound is LDS gather rate"): one 16-byte table entry per (row,
        # sub-quantiser, group of 16 queries), 256 CUs x 256 B/clk x 2.4 GHz for conflict-free ds_read_b128
        lds_peak = 256 * 256 * 2.4                                           # GB/s
        lds_bytes = ((per_rank + 15) // 16) * N_CORPUS * M * 16
        lds_ach = lds_bytes / (scan_ms * 1e-3) / 1e9 if n_l.value else 0.0
        out["adc"] = {
            "metric": "adc_queries_per_sec", "value": round(qps, 1), "unit": "queries/s", "k": k,
            "index": f"{N_CORPUS} x {M} B uint8 uniform codes, resident", "query_batch": nq_batch,
            "batches": args.adc_batches, "ms_per_batch": round(adt / args.adc_batches * 1e3, 2),
            "threshold_head_room_sigmas": ops.ADC_SEL_SLACK,
            "queries_repeated_alone": adc_repeated, "queries_answered_by_the_exact_path": adc_exact,
            "parallelism": f"index replicated, queries split x{world}",
            "parity": "ids and score bits equal the repo's C restatement of Faiss IndexPQ search (tests); Faiss itself is "
                      "not available offline: Faiss-side tie order / last-ulp LUT rounding unpinned",
            "roofline": {"kernel": "adc_screen_q16_kernel<48> (8-bit screening scan: conflict-free ds_read_b128 gathers of "
                                   "16-query byte tables in phases of 16 sub-quantisers, tables double-buffered by LDS-DMA, "
                                   "16x16x64 i8 MFMA accumulation)",
                         "bound": "lds-gather", "achieved": round(lds_ach, 1), "peak": round(lds_peak, 1), "unit": "GB/s",
                         "frac": round(lds_ach / lds_peak, 4), "lds_bytes_per_launch": lds_bytes,
                         "avg_launch_ms": round(scan_ms, 3), "launches_timed": n_l.value,
                         "measured_gather_roof": {
                             "lds_alone_cycles_per_b128_gather_per_cu": 4.05, "with_one_instruction_address_and_mfma_cycles": 4.79,
                             "matrix_pipe_cycles_per_gather_per_cu": 4.0,
                             "note": "tools/ubench_lds_gather.hip (profiles/r03a_ubench_lds_gather.txt, r05a_ubench_lds_gather_perm.txt; "
                                     "not this run): random conflict-free ds_read_b128 gathers with precomputed addresses run at "
                                     "4.05 cycles per wave-instruction per CU = 253 B/clk = the nominal roof used in `frac`; with "
                                     "the screen's one-instruction (v_perm_b32) address and one i8 MFMA per gather 4.79; a "
                                     "v_mfma_i32_16x16x64_i8 occupies a SIMD's matrix unit for 16 cycles = 4.0 per gather per CU, "
                                     "so LDS and matrix pipe saturate together at the ideal; the kernel itself runs at ~9.4 "
                                     "(M = 32, table phases resident: 7.9); on an index with identical table phases the M = 48 "
                                     "kernel without barriers and refills took 6.9 ms against 7.96 (7.6 since the rolling code loads) "
                                     "(profiles/r05p_adc_phase_change.txt, DESIGN.md 4.6, 9.2)"},
                         "hbm_equivalent": {"algorithmic_bytes_per_launch": adc_alg, "achieved_GBs": round(adc_ach, 1),
                                            "note": "N*M code bytes per query (SURVEY 8d) / kernel time: 8 queries share every "
                                                    "code read and tiles are re-read from L2, so this exceeds the HBM peak and is "
                                                    "not a roofline"},
                         "traffic": pmc_traffic("adc_screen_q16_kernel"),
                         "traffic_source": "profiles/pmc_summary.json, not this run"},
        }

    if not args.no_adc:
        # k in {10, 200} at M = 48 and the M = 96 index (BASELINE configs[3] flat leg), one batch each
        sweep = {}
        for kk in (10, 200):
            search(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            index.search(q_all[:nq_batch], kk)
            torch.cuda.synchronize()
            sweep[f"M48_k{kk}"] = round(nq_batch / (time.perf_counter() - t0), 1)
        # survivors of the 8-bit screen per query on the uniform codes (one extra batch, untimed)
        st_u = {}
        index.search_async(q_all[:nq_batch], k, stats=st_u)()
        out["adc"]["screen_survivors_per_query"] = _count_stats(st_u)
        del index
        torch.cuda.empty_cache()

        # ---- SURVEY 8d-D second half: "codes from C (the index build)".  The whole 8 841 823-row corpus of clustered synthetic
        # embeddings is encoded chunk by chunk (nearest codes against Lloyd-refined centroids, the timed part is
        # the assignment) and the same query batches are searched over THAT index: the sampled threshold and the screen's
        # survivor count depend on the code distribution, uniform codes are the easy case.
        if world == 1:
            gcl = torch.Generator(device=dev).manual_seed(20230)
            centers = torch.randn((512, D), device=dev, generator=gcl)

            def corpus_chunk(i, rows):
                gi = torch.Generator(device=dev).manual_seed(20231 + i)
                a = torch.randint(0, centers.shape[0], (rows,), device=dev, generator=gi)
                return (0.7 * centers[a] + 0.5 * torch.randn((rows, D), device=dev, generator=gi)).contiguous()
            x0 = corpus_chunk(0, 1 << 16)
            perm = torch.randperm(1 << 16, device=dev, generator=gcl)[:K]
            Cb = x0[perm].reshape(K, M, D // M).transpose(0, 1).contiguous()
            for _ in range(4):                                   # Lloyd refinement on the first 65 536 rows
                cb = ops.assign_nearest(x0, Cb, torch.uint8)
                sb, nb_ = ops.kmeans_stats(x0, cb)
                ops.kmeans_update_(sb, nb_, Cb)
            del x0
            built = torch.empty((N_CORPUS, M), dtype=torch.uint8, device=dev)
            chunk = 1 << 20
            build_ms = 0.0
            for ci, a0 in enumerate(range(0, N_CORPUS, chunk)):
                rows = min(chunk, N_CORPUS - a0)
                xc_ = corpus_chunk(ci, rows)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                built[a0:a0 + rows] = ops.assign_nearest(xc_, Cb, torch.uint8)
                e1.record()
                torch.cuda.synchronize()
                build_ms += e0.elapsed_time(e1)
                del xc_
            hist0 = torch.bincount(built[:, 0].long(), minlength=K).float()
            index_b = PQIndex(D, M, device=dev)
            index_b.set_centroids(Cb)
            index_b.add_codes(built)
            del built
            index_b.search_async(q_all[:nq_batch], k)()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pend_b = []
            for bi in range(args.adc_batches):
                pend_b.append((index_b.search_async(q_all[bi * nq_batch:(bi + 1) * nq_batch], k), index_b.last_search))
            rep_b = 0
            for fin, pb_ in pend_b:
                fin()
                rep_b += pb_.stats["retried_queries"]
            torch.cuda.synchronize()
            bdt_ = time.perf_counter() - t0
            st_b = {}
            index_b.search_async(q_all[:nq_batch], k, stats=st_b)()
            out["adc"]["index_built_codes"] = {
                "value": round(args.adc_batches * nq_batch / bdt_, 1), "unit": "queries/s", "k": k,
                "ms_per_batch": round(bdt_ / args.adc_batches * 1e3, 2),
                "index": f"{N_CORPUS} x {M} B: nearest codes of clustered synthetic embeddings (512 Gaussian clusters) against "
                         "Lloyd-refined centroids, built on this GPU (SURVEY 8d-C)",
                "screen_survivors_per_query": _count_stats(st_b), "queries_repeated_alone": rep_b,
                "code_histogram_sub_quantiser_0": {"max_over_mean": round(float(hist0.max() / hist0.mean()), 3),
                                                   "empty_codes": int((hist0 == 0).sum())},
                "corpus_build": {"rows": N_CORPUS, "assignment_s": round(build_ms * 1e-3, 4),
                                 "value": round(N_CORPUS / (build_ms * 1e-3), 1), "unit": "vectors/s",
                                 "what": "rc_pq_assign_nearest_fast over the whole corpus in 2^20-row chunks (HIP events around "
                                         "the assignment calls; generating the synthetic rows is not timed)"}}
            del index_b, centers
            torch.cuda.empty_cache()
        M2 = 96
        C96 = torch.randn((M2, K, D // M2), device=dev, generator=torch.Generator(device=dev).manual_seed(20226))
        idx96 = PQIndex(D, M2, device=dev)
        idx96.set_centroids(C96)
        idx96.add_codes(torch.randint(0, 256, (N_CORPUS, M2), dtype=torch.uint8, device=dev, generator=gen))
        idx96.search(q_all[:nq_batch], k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx96.search(q_all[:nq_batch], k)
        torch.cuda.synchronize()
        sweep[f"M96_k{k}"] = round(nq_batch / (time.perf_counter() - t0), 1)
        del idx96
        torch.cuda.empty_cache()
        out["adc"]["other_shapes_queries_per_sec"] = sweep

    # ------------------------------------------------------------------ IVF leg (BASELINE configs[3]: M = 96, nlist = 5000)
    if not args.no_adc and world == 1:
        from repconc_amd.ivf import IVFPQIndex, coarse_assign
        M3, nlist = 96, 5000
        g3 = torch.Generator(device=dev).manual_seed(20227)
        C3 = torch.randn((M3, K, D // M3), device=dev, generator=g3)
        ivf = IVFPQIndex(D, M3, nlist, device=dev)
        ivf.set_centroids(C3)
        ivf.coarse = torch.randn((nlist, D), device=dev, generator=g3)
        # build side: coarse assignment of a 2^18-row chunk of synthetic embeddings on the fp32 matrix cores
        xc = torch.randn((1 << 18, D), device=dev, generator=g3)
        coarse_assign(xc[:4096], ivf.coarse)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cells_c = coarse_assign(xc, ivf.coarse)
        torch.cuda.synchronize()
        cdt = time.perf_counter() - t0
        # the other half of a Lloyd iteration of the coarse quantiser: centroid update (rc_ivf_coarse_update)
        import ctypes as _C
        cu_lib, cu_h = _lib.load(), _lib.handle(dev.index)
        cu_ws = torch.empty((cu_lib.rc_ivf_coarse_update_ws_bytes(1 << 18, nlist),), dtype=torch.uint8, device=dev)
        cu_cent = ivf.coarse.clone()
        cu_assign = cells_c.to(torch.int32)
        cu_s = _C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        def _coarse_update():
            _lib.check(cu_lib.rc_ivf_coarse_update(cu_h, _C.c_void_p(xc.data_ptr()), xc.stride(0), _C.c_void_p(cu_assign.data_ptr()),
                                                   1 << 18, D, nlist, _C.c_void_p(cu_cent.data_ptr()), None, 1234, 0,
                                                   _C.c_void_p(cu_ws.data_ptr()), cu_ws.numel(), cu_s), "rc_ivf_coarse_update", cu_h)
        _coarse_update()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _coarse_update()
        e1.record()
        torch.cuda.synchronize()
        cu_ms = e0.elapsed_time(e1) / 5
        del xc, cu_ws, cu_cent, cu_assign
        # ---- the index ("IVF measured as retrieval, not as a scan"): a clustered synthetic corpus (512 Gaussian
        # clusters, weaker than the flat leg's: 0.3 centre + 0.5 noise; 8 841 823 rows, generated chunk by chunk), PQ centroids Lloyd-refined on its first 65 536
        # rows, the coarse quantiser TRAINED on its first 2^20 rows (10 Lloyd iterations of rc_ivf_coarse_assign /
        # rc_ivf_coarse_update, Faiss's default for an IVF coarse quantiser), every row coded (nearest codes) and sent to its
        # L2-nearest cell.  Queries = noisy copies of corpus rows: the copied row is the query's one relevant passage.
        gcl3 = torch.Generator(device=dev).manual_seed(20230)
        centers3 = torch.randn((512, D), device=dev, generator=gcl3)

        def corpus_chunk3(i, rows):
            gi = torch.Generator(device=dev).manual_seed(20231 + i)
            a_ = torch.randint(0, centers3.shape[0], (rows,), device=dev, generator=gi)
            return (0.3 * centers3[a_] + 0.5 * torch.randn((rows, D), device=dev, generator=gi)).contiguous()
        chunk3 = 1 << 20
        x0 = corpus_chunk3(0, chunk3)
        perm3 = torch.randperm(1 << 16, device=dev, generator=gcl3)[:K]
        C3 = x0[perm3].reshape(K, M3, D // M3).transpose(0, 1).contiguous()
        for _ in range(4):
            cb3 = ops.assign_nearest(x0[:1 << 16], C3, torch.uint8)
            sb3, nb3 = ops.kmeans_stats(x0[:1 << 16], cb3)
            ops.kmeans_update_(sb3, nb3, C3)
        ivf.set_centroids(C3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        from repconc_amd.ivf import coarse_kmeans
        ivf.coarse = coarse_kmeans(x0, nlist, iters=10)
        torch.cuda.synchronize()
        coarse_train_s = time.perf_counter() - t0
        nq_dev = 6980
        rel = torch.from_numpy(np.sort(np.random.default_rng(20232).permutation(chunk3)[:nq_dev]).copy()).to(dev)   # rows of chunk 0
        q_dev = (x0[rel] + 2.0 * torch.randn((nq_dev, D), device=dev, generator=gcl3)).contiguous()
        codes3 = torch.empty((N_CORPUS, M3), dtype=torch.uint8, device=dev)
        cells3 = torch.empty((N_CORPUS,), dtype=torch.int64, device=dev)
        t0 = time.perf_counter()
        for ci, a0 in enumerate(range(0, N_CORPUS, chunk3)):
            rows = min(chunk3, N_CORPUS - a0)
            xc_ = x0 if ci == 0 else corpus_chunk3(ci, rows)
            codes3[a0:a0 + rows] = ops.assign_nearest(xc_[:rows], C3, torch.uint8)
            cells3[a0:a0 + rows] = coarse_assign(xc_[:rows], ivf.coarse)
            del xc_
        torch.cuda.synchronize()
        ivf_build_s = time.perf_counter() - t0
        del x0
        ivf.set_lists(codes3, cells3)
        cell_sizes = (ivf.list_off[1:] - ivf.list_off[:-1]).float()
        del codes3, cells3
        qi = q_dev[:nq_batch]
        sweep3 = {}
        for nprobe in (8, 32, 128):
            for _ in range(2):                                   # the first calls size the allocator's workspace blocks
                ivf.search(qi, k, nprobe)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for bi in range(args.adc_batches):                   # the 1200-query batches of the flat leg
                b0 = min(bi * nq_batch, nq_dev - nq_batch)
                ivf.search(q_dev[b0:b0 + nq_batch], k, nprobe)
            torch.cuda.synchronize()
            sweep3[f"nprobe{nprobe}"] = round(args.adc_batches * nq_batch / (time.perf_counter() - t0), 1)
        # the evaluation's own call (batch_search hands a list-centric index every query it has): the 6 980 MS MARCO dev queries
        # in ONE call — a probed cell is scanned once for all the queries that probe it — with the retrieval quality of every
        # nprobe against the flat search of the same index: recall@10 / recall@1000 (overlap of the id sets) and MRR@10 of the
        # planted relevant passage (`north_star`: MRR@10 within +-0.001)
        flat3 = PQIndex(D, M3, device=dev)
        flat3.set_centroids(C3)
        flat3.add_codes(ivf.codes)
        f_ids = torch.cat([ivf.ids[flat3.search(q_dev[i:i + nq_batch], k)[1]] for i in range(0, nq_dev, nq_batch)])

        def mrr10(ids_):
            hit = (ids_[:, :10] == rel[:, None])
            rr = (hit.float() / torch.arange(1, 11, device=dev, dtype=torch.float32)[None, :]).sum(1)
            return float(rr.mean())

        def overlap(a_, b_, kk):
            a_, b_ = a_[:, :kk].contiguous(), b_[:, :kk].contiguous()
            return float((a_[:, :, None] == b_[:, None, :]).any(2).float().mean()) if kk <= 10 else \
                float(torch.stack([torch.isin(a_[i], b_[i]).float().mean() for i in range(0, a_.shape[0], 7)]).mean())
        mrr_flat = mrr10(f_ids)
        sweep3_all, retrieval = {}, {}
        for nprobe in (8, 32, 128, 512, 2048):
            ivf.search(q_dev, k, nprobe)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                _, i_ids = ivf.search(q_dev, k, nprobe)
            torch.cuda.synchronize()
            sweep3_all[f"nprobe{nprobe}"] = round(3 * nq_dev / (time.perf_counter() - t0), 1)
            retrieval[f"nprobe{nprobe}"] = {"recall_at_10_vs_flat": round(overlap(i_ids, f_ids, 10), 4),
                                            f"recall_at_{k}_vs_flat": round(overlap(i_ids, f_ids, k), 4),
                                            "mrr_at_10": round(mrr10(i_ids), 5),
                                            "mrr_at_10_minus_flat": round(mrr10(i_ids) - mrr_flat, 5)}
        within = [int(n_[6:]) for n_, r_ in retrieval.items() if abs(r_["mrr_at_10_minus_flat"]) <= 0.001]
        retrieval_summary = {"mrr_at_10_flat": round(mrr_flat, 5),
                             "smallest_nprobe_with_mrr_within_0.001_of_flat": min(within) if within else None,
                             "mrr_at_10": {n_: r_["mrr_at_10"] for n_, r_ in retrieval.items()},
                             "recall_at_10": {n_: r_["recall_at_10_vs_flat"] for n_, r_ in retrieval.items()}}
        del f_ids
        # nprobe = nlist scans every row: must equal the flat search (checked on 64 queries)
        fs, fi = flat3.search(qi[:64], 100)
        s3, i3 = ivf.search(qi[:64], 100, nlist)
        same3 = bool(torch.equal(fs, s3) and float((ivf.ids[fi] == i3).float().mean()) > 0.999)   # ids may swap inside exact score ties
        rows128 = float(cell_sizes[ivf.probe(qi, 128, ordered=False).long()].sum(1).mean())   # probed rows per query, measured
        t128 = nq_batch / sweep3["nprobe128"]
        # the screen kernel alone (HIP events around its launch), nprobe = 128: against the measured LDS gather roof
        lib.rc_profile_enable(h, 1)
        ivf.search(qi, k, 128)
        torch.cuda.synchronize()
        lib.rc_profile_enable(h, 0)
        lib.rc_profile_collect(h, _lib.PROF_ADC_SCAN, ctypes.byref(n_l), ctypes.byref(ms_l))
        ivf_scan_ms = ms_l.value / max(n_l.value, 1)
        ivf_gather = nq_batch * rows128 * M3 / (ivf_scan_ms * 1e-3) / 1e9 if n_l.value else 0.0   # 1 B per (row, m, query)
        out["ivf"] = {
            "metric": "ivf_adc_queries_per_sec", "unit": "queries/s", "k": k, "nlist": nlist, "M": M3,
            "index": f"{N_CORPUS} x {M3} B: nearest codes of a clustered synthetic corpus (512 Gaussian clusters, 0.3 centre + 0.5 noise) in {nlist} cells of "
                     "a coarse quantiser trained on its first 2^20 rows (10 Lloyd iterations), rows in their L2-nearest cell, no "
                     "residual coding",
            "cells": {"rows_min": int(cell_sizes.min()), "rows_max": int(cell_sizes.max()),
                      "rows_mean": round(float(cell_sizes.mean()), 1), "empty": int((cell_sizes == 0).sum())},
            "build": {"coarse_training_s": round(coarse_train_s, 3), "coding_and_cell_assignment_s": round(ivf_build_s, 3),
                      "what": "coarse_kmeans on 2^20 rows; then nearest codes + nearest cell of all rows in 2^20-row chunks "
                              "(wall time incl. generating the synthetic rows)"},
            "queries": "6980 noisy copies (sigma 2.0 per coordinate: |noise| = 3.4 x |row|) of corpus rows; the copied row is the one relevant passage",
            "queries_per_sec": sweep3_all, "queries_per_call": 6980,
            "retrieval": retrieval, "retrieval_summary": retrieval_summary,
            "queries_per_sec_1200_query_calls": sweep3, "nprobe_equals_nlist_matches_flat_search": same3,
            "coarse_assign": {"value": round((1 << 18) / cdt, 1), "unit": "vectors/s", "rows": 1 << 18,
                              "roofline": {"kernel": "ivf_coarse_assign_kernel (v_mfma_f32_32x32x2_f32, fused argmin)",
                                           "bound": "mfma", "achieved": round(2.0 * (1 << 18) * nlist * D / cdt / 1e12, 2),
                                           "peak": 157.3, "unit": "TFLOP/s",
                                           "frac": round(2.0 * (1 << 18) * nlist * D / cdt / 1e12 / 157.3, 4)}},
            "coarse_update": {"ms": round(cu_ms, 4), "rows": 1 << 18, "value": round((1 << 18) / cu_ms * 1e3, 1), "unit": "vectors/s",
                              "roofline": {"kernel": "ivfc_cell_mean_kernel (+ stable counting sort: tile histograms, scan, scatter)",
                                           "bound": "hbm", "achieved": round((1 << 18) * D * 4 / cu_ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": round((1 << 18) * D * 4 / cu_ms / 1e6 / HBM_PEAK_GBS, 4)},
                              "note": "centroid update of one Lloyd iteration of the coarse quantiser, 4 D bytes per row; the "
                                      "whole rc_ivf_coarse_update call (4 kernels), HIP events around 5 calls"},
            "roofline": {"kernel": "ivfs_screen16_kernel<96, 4> (round 6; list-centric: one persistent block per CU walks (cell, <= 16 probing "
                                   "queries) tasks; 64 KiB table phases of 16 sub-quantisers x 16 queries in two LDS buffers, four loader "
                                   "waves request a phase one stage ahead and byte-transpose it into the other buffer while twelve waves "
                                   "gather; conflict-free ds_read_b128 gathers + i8 MFMA accumulation; survivors to per-wave streams); "
                                   "calls with fewer than 9 queries per probed cell run the 8-query form (ivfs_screen_kernel, "
                                   "ds_read_b64), nprobe < 6 the per-query scan",
                         "bound": "lds-gather", "achieved": round(ivf_gather, 1), "peak": round(256 * 256 * 2.4, 1),
                         "unit": "GB/s", "frac": round(ivf_gather / (256 * 256 * 2.4), 4),
                         "screen_kernel_ms": round(ivf_scan_ms, 3), "nprobe": 128,
                         "probed_rows_per_query": round(rows128, 1),
                         "note": "nprobe = 128, 1200-query call, the screen kernel alone (HIP events): one table byte per (probed row, "
                                 "sub-quantiser, query) against the nominal conflict-free gather rate (256 CUs x 256 B/clk x 2.4 "
                                 "GHz).  The LDS pipe also takes the tables: a task of 16 queries writes 16 x 96 x 256 B = 384 KiB "
                                 "(ds_write_b128, ~80 B/clk) for ~1770 rows x 96 x 16 B of gathers - 4.9 k of every 17.6 k LDS cycles "
                                 "of a task are table stores -, and the sixteen waves of a block meet at one barrier per 64 KiB phase "
                                 "(six per task): gathers + stores at their nominal rates are 0.4 of the measured time "
                                 "(DESIGN.md 4.7)",
                         "whole_search_equivalent_code_GBs": round(nq_batch * rows128 * M3 / t128 / 1e9, 1)}}
        del ivf, flat3, centers3
        torch.cuda.empty_cache()
        # ---- the scan-rate reference of rounds 2-5, kept for comparability: the same sizes with uniformly random codes in
        # uniformly filled cells and Gaussian queries (no retrieval quality to speak of: top-10 overlap with flat = nprobe / nlist)
        gu = torch.Generator(device=dev).manual_seed(20227)
        ivf_u = IVFPQIndex(D, M3, nlist, device=dev)
        ivf_u.set_centroids(torch.randn((M3, K, D // M3), device=dev, generator=gu))
        ivf_u.coarse = torch.randn((nlist, D), device=dev, generator=gu)
        codes_u = torch.randint(0, 256, (N_CORPUS, M3), dtype=torch.uint8, device=dev, generator=gu)
        ivf_u.set_lists(codes_u, torch.randint(0, nlist, (N_CORPUS,), device=dev, generator=gu))
        del codes_u
        q_u = torch.randn((nq_dev, D), device=dev, generator=gu)
        uni_all, uni_1200 = {}, {}
        for nprobe in (8, 32, 128):
            for qs_, dst in ((q_u, uni_all), (q_u[:nq_batch], uni_1200)):
                ivf_u.search(qs_, k, nprobe)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    ivf_u.search(qs_, k, nprobe)
                torch.cuda.synchronize()
                dst[f"nprobe{nprobe}"] = round(3 * qs_.shape[0] / (time.perf_counter() - t0), 1)
        lib.rc_profile_enable(h, 1)
        ivf_u.search(q_u[:nq_batch], k, 128)
        torch.cuda.synchronize()
        lib.rc_profile_enable(h, 0)
        lib.rc_profile_collect(h, _lib.PROF_ADC_SCAN, ctypes.byref(n_l), ctypes.byref(ms_l))
        u_ms = ms_l.value / max(n_l.value, 1)
        u_rows = N_CORPUS * 128 / nlist
        u_gather = nq_batch * u_rows * M3 / (u_ms * 1e-3) / 1e9 if n_l.value else 0.0
        out["ivf"]["uniform_cells"] = {
            "index": f"{N_CORPUS} x {M3} B uniform codes in {nlist} uniformly filled cells (the IVF leg of rounds 2-5)",
            "queries_per_sec": uni_all, "queries_per_call": nq_dev, "queries_per_sec_1200_query_calls": uni_1200,
            "roofline": {"bound": "lds-gather", "achieved": round(u_gather, 1), "peak": round(256 * 256 * 2.4, 1), "unit": "GB/s",
                         "frac": round(u_gather / (256 * 256 * 2.4), 4), "screen_kernel_ms": round(u_ms, 3), "nprobe": 128,
                         "note": "the screen kernel of a 1200-query call at nprobe 128, same definition as rounds 3-5 (0.164 in round 5)"}}
        del ivf_u, q_u, q_dev
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ index-build leg (nearest codes, a-1/a-5)
    if not args.no_adc:
        nb = 1 << 20                                           # rows per rank per pass (a 1 M-passage encode chunk)
        xb = torch.randn((nb, D), device=dev, generator=torch.Generator(device=dev).manual_seed(20225 + rank))
        ib = {}
        for method in ("exact", "mfma"):
            ops.assign_nearest(xb, C, torch.uint8, method=method)      # warm-up at full size (PMC means stay per-pass)
            barrier()
            lib.rc_profile_enable(h, 1)
            t0 = time.perf_counter()
            st = {}
            for _ in range(3):
                nc = ops.assign_nearest(xb, C, torch.uint8, method=method, stats=st)
            barrier()
            bdt = max_over_ranks(time.perf_counter() - t0)
            lib.rc_profile_enable(h, 0)
            lib.rc_profile_collect(h, _lib.PROF_ASSIGN_NEAREST, ctypes.byref(n_l), ctypes.byref(ms_l))
            ib[method] = (3 * nb * world / bdt, ms_l.value / max(n_l.value, 1), st.get("doubtful", 0), nc)
        same = bool(torch.equal(ib["exact"][3], ib["mfma"][3]))
        bytes_per_vec = D * 4 + M
        out["index_build"] = {
            "metric": "nearest_code_assignments_per_sec", "value": round(ib["mfma"][0], 1), "unit": "vectors/s",
            "rows_per_pass_per_gpu": nb, "kernel_ms_per_pass": round(ib["mfma"][1], 3),
            "method": "split-bf16 MFMA screen + exact fp32 rescoring of the doubtful pairs (csrc/pq_assign_mfma.hip)",
            "doubtful_pairs_per_pass": ib["mfma"][2], "codes_identical_to_exact_kernel": same,
            "exact_kernel": {"value": round(ib["exact"][0], 1), "kernel_ms_per_pass": round(ib["exact"][1], 3)},
            "roofline": {"kernel": "assign_mfma_kernel<16> + assign_redo_kernel<16>", "bound": "hbm",
                         "achieved": round(nb * bytes_per_vec / (ib["mfma"][1] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(nb * bytes_per_vec / (ib["mfma"][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic("assign_mfma_kernel"),
                         "traffic_source": "profiles/pmc_summary.json, not this run",
                         "algorithmic_bytes_per_launch": nb * bytes_per_vec,
                         "valu_issue_bound": _valu_bound(nb, M, K, ib["mfma"][1]),
                         "note": "far from the HBM roof by construction: 256 candidate distances per (row, sub-quantiser), "
                                 "2.25 VALU instructions each in the pair-folded min/second-min epilogue, the bf16 MFMAs "
                                 "underneath; after round 2 neither pipe is saturated (VALU ~64 %, matrix pipe ~37 %, waves "
                                 "wait on LDS operand reads and the per-sub-quantiser barrier; DESIGN.md §4.4, §9.4)"},
        }
        del xb

    # ------------------------------------------------------------------ warm-up leg (OPQ + PQ training, a-12)
    if world == 1 and not args.no_adc and not args.no_opq:
        from repconc_amd.train.run_warmup import MAX_TRAIN_POINTS, _xt_y, train_opq, train_pq
        gw = torch.Generator(device=dev).manual_seed(20226)
        xt = torch.randn((MAX_TRAIN_POINTS, D), device=dev, generator=gw)
        xt = (xt @ (torch.randn((D, D), device=dev, generator=gw) / D ** 0.5)).contiguous()    # correlated dimensions
        train_pq(xt, M, 1)                                        # library / kernel warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Cw, mse_pq = train_pq(xt, M, 25)
        torch.cuda.synchronize()
        t_pq = time.perf_counter() - t0
        # untimed: three short rounds on the same shapes load what a first call pays once per process (the device QR of the
        # initial rotation, the libraries' fp64 / transposed GEMM kernels, the capture machinery) - 0.35 s cold, 0.30 s after
        train_opq(xt, M, n_outer=3, n_pq_first=2, n_pq=1)
        torch.cuda.synchronize()
        hist = []
        t0 = time.perf_counter()
        Rw = train_opq(xt, M, n_outer=50, n_pq_first=40, n_pq=4, history=hist)
        torch.cuda.synchronize()
        t_opq = time.perf_counter() - t0
        xr = (xt @ Rw).contiguous()
        t0 = time.perf_counter()
        Cw, mse_opq = train_pq(xr, M, 25)
        torch.cuda.synchronize()
        t_pq2 = time.perf_counter() - t0
        Pm = (xt.T @ xt).double()
        t0 = time.perf_counter()
        torch.linalg.svd(Pm)
        torch.cuda.synchronize()
        t_svd = time.perf_counter() - t0
        from repconc_amd.train.run_warmup import procrustes_rotation
        t0 = time.perf_counter()
        procrustes_rotation(Pm)
        torch.cuda.synchronize()
        t_polar = time.perf_counter() - t0

        # what one OPQ round is made of (HIP-event time of each part, device-side only) next to the round as it runs
        def ev_ms(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        codes_w = ops.assign_nearest(xr, Cw, torch.uint8)
        xrec_w = ops.decode_raw(codes_w, Cw)
        Pw = (xt.T @ xrec_w).double()

        def lloyd():
            cw = ops.assign_nearest(xr, Cw, torch.uint8)
            sw, nw = ops.kmeans_stats(xr, cw)
            ops.kmeans_update_(sw, nw, Cw.clone())
        parts = {
            "rotate_gemm_ms": ev_ms(lambda: (xt @ Rw).contiguous()),
            "lloyd_iteration_ms": ev_ms(lloyd),
            "assign_nearest_ms": ev_ms(lambda: ops.assign_nearest(xr, Cw, torch.uint8)),
            "decode_ms": ev_ms(lambda: ops.decode_raw(codes_w, Cw)),
            "xT_xrec_gemm_fp64cast_ms": ev_ms(lambda: _xt_y(xt, xrec_w)),       # 16 row slices as one batched product
            "procrustes_ms": ev_ms(lambda: procrustes_rotation(Pw), reps=3),
        }
        parts = {kk: round(v, 3) for kk, v in parts.items()}
        # rounds >= 1 (one graph replay each): 4 Lloyd iterations, ONE final assignment + decode (round 3 ran them twice)
        device_sum = (parts["rotate_gemm_ms"] + 4 * parts["lloyd_iteration_ms"] + parts["assign_nearest_ms"] + parts["decode_ms"]
                      + parts["xT_xrec_gemm_fp64cast_ms"] + parts["procrustes_ms"])
        parts["device_sum_per_round_ms"] = round(device_sum, 3)
        parts["measured_per_round_ms"] = round(t_opq / 50 * 1e3, 3)
        parts["note"] = ("a round = rotate, 4 Lloyd iterations (assignment, statistics, update, empty-cluster rule: no host "
                         "synchronisation), error, assignment, decode, x^T x_rec, Procrustes (optimally scaled Newton-Schulz on a fixed "
                         "36-step schedule, orthogonality check left on the device): no host synchronisation in a round, and since "
                         "round 4 the WHOLE round (~70 launches of the package's kernels, ~80 library GEMMs, ~40 element-wise ops) is "
                         "one hipGraph replay on static buffers (run_warmup._RoundGraph; bit-identical to the eager rounds), so the "
                         "host's launch rate no longer sets the pace; measured - device sum = gaps between the ~200 graph nodes + "
                         "round 0 (40 Lloyd iterations, eager) + round 1 (eager, then captured)")
        del codes_w, xrec_w, Pw
        out["opq_pq_warmup"] = {
            "metric": "opq_pq_training_seconds", "value": round(t_opq + t_pq2, 3), "unit": "s", "higher_is_better": False,
            "train_rows": MAX_TRAIN_POINTS, "M": M,
            "opq_50_rounds_s": round(t_opq, 3), "final_pq_25_lloyd_iterations_s": round(t_pq2, 3),
            "lloyd_iteration_ms": round(t_pq / 26 * 1e3, 3),
            "procrustes_newton_schulz_s": round(t_polar, 4), "procrustes_library_svd_fp64_s": round(t_svd, 3),
            "mse_pq_without_rotation": round(mse_pq, 5), "mse_after_opq": round(mse_opq, 5),
            "mse_opq_rounds_first_last": [round(hist[0], 5), round(hist[-1], 5)],
            "rotation_orthogonality_error": float((Rw @ Rw.T - torch.eye(D, device=dev)).abs().max()),
            "opq_round_breakdown": parts,
            "note": "the whole training of run_warmup.py:92-113 at Faiss-default sizes (65 536 training rows; 50 OPQ rounds "
                    "with 40 + 49 x 4 Lloyd iterations, then 25 Lloyd iterations on the rotated rows), nothing projected: "
                    "assignment / statistics / update are the HIP kernels, the 768-wide GEMMs are library calls, the "
                    "Procrustes step is a GEMM-only, optimally scaled Newton-Schulz polar iteration in fp64 (the library SVD beside it)",
        }
        del xr, Pm
        del xt, Cw, Rw

    # ------------------------------------------------------------------ k-means sufficient statistics (GPU leg)
    # at the sizes the path runs it on: the 65 536 training rows of the warm-up (run_warmup.py:92-113) and a 2^20-row corpus
    # chunk (corpus-sharded k-means, BASELINE configs[2]); the CPU port is timed beside it in the cpu_baseline block
    if world == 1:
        ks_sizes = {}
        gk = torch.Generator(device=dev).manual_seed(20228)
        for rows in (1 << 16, 1 << 20):
            xg = torch.randn((rows, D), device=dev, generator=gk)
            cg = torch.randint(0, 256, (rows, M), dtype=torch.uint8, device=dev, generator=gk)
            for _ in range(3):
                ops.kmeans_stats(xg, cg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.kmeans_stats(xg, cg)
            e1.record()
            torch.cuda.synchronize()
            kg = e0.elapsed_time(e1) / 10 * 1e-3
            ks_sizes[rows] = {"rows": rows, "ms": round(kg * 1e3, 4), "value": round(rows / kg, 1), "unit": "vectors/s",
                              "roofline": {"kernel": "kmeans_stats_px_kernel<0> (+ <1>: the finish, <2>: the idle repeat pass; integer split of "
                                                     "the fp32 bits, 64-bit LDS accumulators, 128-byte row pieces; the whole "
                                                     "rc_kmeans_stats call is timed)",
                                           "bound": "hbm", "achieved": round(rows * (D * 4 + M) / kg / 1e9, 1),
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(rows * (D * 4 + M) / kg / 1e9 / HBM_PEAK_GBS, 4),
                                           "algorithmic_bytes_per_launch": rows * (D * 4 + M)}}
            del xg, cg
        big = ks_sizes[1 << 20]
        out["kmeans_stats"] = {"metric": "kmeans_sufficient_statistics_vectors_per_sec", "value": big["value"],
                               "unit": "vectors/s", "rows": big["rows"], "ms": big["ms"], "roofline": big["roofline"],
                               "warmup_size_65536": ks_sizes[1 << 16],
                               "note": "4 D + M bytes per vector (SURVEY 8d), HIP events around 10 calls",
                               "warmup_size_floor": "65 536 rows = 204 MB: three launches (profiles/r05h_kmeans_phases.txt) - A rows -> "
                                                    "per-strip partials 47 us (the rows at 5-7 TB/s 29-39 us, + 34 MB of 64-bit "
                                                    "partial accumulators written: 8 us), B the finish on every CU 10.8 us (34 MB "
                                                    "read back, integer sum over the strips, one rounding), C closes the call 7.4 us; "
                                                    "a one-launch form with the finish in the last-arriving block measured 103-108 us "
                                                    "(DESIGN 9.3).  The rows alone at the achievable 6.3 TB/s are 32 us = 0.49 of the "
                                                    "call: the fixed ~27 us of partials + finish + close is what a 204 MB input "
                                                    "cannot amortise (2^20 rows: 0.52-0.61)"}

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1)
    if world == 1 and not args.no_cpu:
        from oracle import c_oracle, torch_port
        # threads = what the host really gives this process: the GPU boxes of the pool show 256 logical CPUs but run the
        # container under a cgroup quota of 16 (cpu.max "1600000 100000"): 128 OpenMP threads time-sliced on 16 CPUs is what made
        # the round-4 baselines scale 5-7 x on "128 cores".  The quota (rounded up) bounds the thread count; `cores` states it.
        cores = c_oracle.num_threads()
        quota = _host_cpus().get("quota_cpus")
        if quota:
            cores = max(1, min(cores, int(quota)))
        c_oracle.set_num_threads(cores)
        Bs = 32768                                           # ~10 s of host time on the pool's 16-CPU quota
        xs = np.random.default_rng(20224).standard_normal((Bs, D), dtype=np.float32)
        t0 = time.perf_counter()
        cpu_codes, _ = c_oracle.quantize(xs, cent, True, EPS, ITERS)
        cdt = time.perf_counter() - t0
        got, _ = ops.assign_sinkhorn(torch.from_numpy(xs).to(dev), C, EPS, ITERS, torch.uint8)
        agree = bool(np.array_equal(got.cpu().numpy(), cpu_codes))
        port = {"value": round(Bs / cdt, 1), "unit": "vectors/s", "cores": cores, "kind": "port",
                "sample": f"one {Bs}x768 batch, M=48, eps=0.003, 100 iterations: oracle/pq_oracle.c (OpenMP C port of the "
                          f"reference's fp32 distance table + in-place fp64 Sinkhorn), {cdt:.1f} s; GPU codes identical: {agree}"}
        # BASELINE.md 4.1 / SURVEY 8d: the reference's algorithmic shape on torch-CPU (what RepCONC.quantize itself does on a CPU
        # box): materialised [M, B, K, dsub] fp32 scratch, fp64 Sinkhorn on the whole [M, K, B] matrix, 4096-row batches, torch's
        # intra-op thread pool on all cores.  Two batches after a small warm-up call.
        torch.set_num_threads(cores)
        ct = torch.from_numpy(cent)
        torch_port.quantize(torch.from_numpy(xs[:256]), ct, True, EPS, 3)
        t0 = time.perf_counter()
        tcodes = [torch_port.quantize(torch.from_numpy(xs[a:a + 4096]), ct, True, EPS, ITERS) for a in (0, 4096, 8192)]
        tdt = time.perf_counter() - t0
        got_t = [ops.assign_sinkhorn(torch.from_numpy(xs[a:a + 4096]).to(dev), C, EPS, ITERS, torch.uint8)[0] for a in (0, 4096, 8192)]
        agree_t = all(bool(np.array_equal(g_.cpu().numpy(), t_.numpy().astype(np.uint8))) for g_, t_ in zip(got_t, tcodes))
        tport = {"value": round(3 * 4096 / tdt, 1), "unit": "vectors/s", "cores": cores, "kind": "torch",
                 "sample": f"three 4096x768 batches, M=48, eps=0.003, 100 iterations: oracle/torch_port.py (torch-CPU restatement in "
                           f"the reference's own formulation, {torch.get_num_threads()} intra-op threads), {tdt:.1f} s; GPU codes "
                           f"identical: {agree_t}"}
        best = port if port["value"] >= tport["value"] else tport
        out["cpu_baseline"] = dict(best, cpu=_cpu_model(), host=_host_cpus(),
                                   candidates={"port": port, "torch": tport},
                                   faiss_available=_faiss_or_none() is not None,
                                   note="the faster of the two CPU restatements of the reference on this host's cores is the "
                                        "stated baseline (Faiss plays no part in the constrained assignment)")
        out["speedup_vs_cpu_baseline"] = round(value / best["value"], 1)
        # the same port on ONE thread (1024-row batch: ~10 s)
        c_oracle.set_num_threads(1)
        t0 = time.perf_counter()
        c_oracle.quantize(xs[:1024], cent, True, EPS, ITERS)
        c1 = time.perf_counter() - t0
        out["cpu_baseline_single_thread"] = {"value": round(1024 / c1, 1), "unit": "vectors/s", "cores": 1, "kind": "port",
                                             "sample": f"one 1024x768 batch, same port, {c1:.1f} s"}
        out["cpu_baseline"]["scaling_vs_one_thread"] = round(best["value"] / (1024 / c1), 1)
        # nearest-code assignment (index build) and k-means sufficient statistics on the host
        t0 = time.perf_counter()
        near1, _ = c_oracle.quantize(xs[:512], cent, False)
        n1 = time.perf_counter() - t0
        c_oracle.set_num_threads(cores)
        t0 = time.perf_counter()
        near_c, _ = c_oracle.quantize(xs, cent, False)
        nall = time.perf_counter() - t0
        from oracle import pq_oracle
        t0 = time.perf_counter()
        pq_oracle.kmeans_stats(xs, near_c, M)
        ks = time.perf_counter() - t0
        if "index_build" in out:
            out["index_build"]["cpu_baseline"] = {
                "value": round(Bs / nall, 1), "unit": "vectors/s", "cores": cores, "kind": "port",
                "sample": f"{Bs} rows, exact fp32 nearest codes, oracle/pq_oracle.c ({nall:.2f} s); one thread: "
                          f"{512 / n1:.0f} vectors/s (512 rows, {n1:.2f} s)"}
            out["index_build"]["speedup_vs_cpu_baseline"] = round(out["index_build"]["value"] / (Bs / nall), 1)
        if "kmeans_stats" in out:
            out["kmeans_stats"]["cpu_baseline"] = {"value": round(Bs / ks, 1), "unit": "vectors/s", "cores": 1, "kind": "port",
                                                   "sample": f"{Bs} rows, numpy restatement oracle/pq_oracle.py ({ks:.2f} s)"}
        if not args.no_adc:
            # CPU comparator of the search (SURVEY 8d / BASELINE.md 4.2): Faiss itself when the box has it, else the C port.
            # The code array is copied with a parallel first touch (its pages on every memory node of the host: the r4 line
            # read 42 GB/s on 128 threads from ONE node), and the port is timed in two forms: Faiss's own loop — one query per
            # thread, each streaming the whole index — and the cache-blocked form (row tiles outside, queries inside: a tile is
            # read from DRAM once per batch of queries); identical results, the faster one is the stated baseline.
            sl = c_oracle.first_touch_copy(index_codes.cpu().numpy())
            nq_t = min(64 * cores, int(q_all.shape[0]))         # 64 queries per thread in the cache-blocked form (~5 s)
            qc = q_all[:nq_t].cpu().numpy()
            t0 = time.perf_counter()
            cpu_s, cpu_i = c_oracle.adc_search(sl, cent, qc, k, tile=0)
            adt_t = time.perf_counter() - t0
            nq_c = min(32 * cores, nq_t)                         # Faiss's loop: 32 queries per thread (each scans 424 MB; ~3 s)
            t0 = time.perf_counter()
            qp_s, qp_i = c_oracle.adc_search(sl, cent, qc[:nq_c], k)
            adt_c = time.perf_counter() - t0
            forms_equal = bool(np.array_equal(qp_i, cpu_i[:nq_c]) and np.array_equal(qp_s.view(np.uint32), cpu_s[:nq_c].view(np.uint32)))
            cands = {"port_cache_blocked": {"value": round(nq_t / adt_t, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                                            "sample": f"{nq_t} queries over the whole {N_CORPUS}-row index ({adt_t:.1f} s): "
                                                      "oracle/pq_oracle.c orc_adc_search_tiled (16384-row tiles, per-query LUT + "
                                                      "size-k heap, queries spread over the threads inside a tile)"},
                     "port_faiss_loop": {"value": round(nq_c / adt_c, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                                         "sample": f"{nq_c} queries ({adt_c:.1f} s): orc_adc_search (Faiss's IndexPQ loop: one "
                                                   "query per thread, linear scan of the whole index four rows at a time)"}}
            faiss = _faiss_or_none()
            if faiss is not None:
                try:
                    fidx = faiss.IndexPQ(D, M, 8, faiss.METRIC_INNER_PRODUCT)
                    faiss.copy_array_to_vector(np.ascontiguousarray(cent).ravel(), fidx.pq.centroids)
                    fidx.is_trained = True
                    faiss.copy_array_to_vector(sl.ravel(), fidx.codes)
                    fidx.ntotal = N_CORPUS
                    faiss.omp_set_num_threads(cores)
                    t0 = time.perf_counter()
                    fidx.search(qc[:nq_c], k)
                    fdt = time.perf_counter() - t0
                    cands["faiss"] = {"value": round(nq_c / fdt, 2), "unit": "queries/s", "cores": cores, "kind": "faiss",
                                      "sample": f"{nq_c} queries ({fdt:.1f} s): faiss.IndexPQ.search, faiss {faiss.__version__}"}
                except Exception as e:                       # an unexpected Faiss build: say so, keep the port
                    cands["faiss_error"] = f"{type(e).__name__}: {e}"
            bestk = "faiss" if "faiss" in cands else max(("port_cache_blocked", "port_faiss_loop"), key=lambda c_: cands[c_]["value"])
            qps_c = cands[bestk]["value"]
            nq_c = nq_t
            # the same queries on the GPU: ids and score bits of the CPU restatement over the WHOLE 8.84 M-row index
            gpu_s, gpu_i = ops.adc_search(index_codes, C, q_all[:nq_c], k)
            out["adc"]["gpu_ids_identical"] = bool(np.array_equal(gpu_i.cpu().numpy(), cpu_i))
            out["adc"]["gpu_score_bits_identical"] = bool(np.array_equal(gpu_s.cpu().numpy().view(np.uint32), cpu_s.view(np.uint32)))
            out["adc"]["checked_against_cpu_port"] = f"{nq_c} queries x top-{k} over the whole index"
            del gpu_s, gpu_i
            out["adc"]["cpu_baseline"] = dict(cands[bestk], candidates=cands, both_port_forms_identical=forms_equal,
                                              code_array="copied with a parallel first touch (pages on every memory node)")
            out["adc"]["speedup_vs_cpu_baseline"] = round(out["adc"]["value"] / qps_c, 1)
            # the reference's own evaluation default is ONE Faiss thread (EvalArguments.threads = 1,
            # evaluate_repconc.py:37; run_repconc_eval.py:149): 3 queries over the whole index on one core
            c_oracle.set_num_threads(1)
            t0 = time.perf_counter()
            c_oracle.adc_search(sl, cent, qc[:3], k)
            adt_1 = time.perf_counter() - t0
            c_oracle.set_num_threads(cores)
            out["adc"]["cpu_baseline_single_thread"] = {
                "value": round(3 / adt_1, 3), "unit": "queries/s", "cores": 1, "kind": "port",
                "sample": f"3 queries over the whole index on one thread ({adt_1:.1f} s) - the reference's eval default "
                          "threads=1"}

    # ------------------------------------------------------------------ roofline.traffic measured by THIS run (N = 1)
    under_profiler = any("rocprof" in os.environ.get(v_, "") for v_ in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES"))   # no nesting
    if world == 1 and not args.no_traffic and B == B_GLOBAL and not under_profiler:
        torch.cuda.empty_cache()
        live = measure_sweep_traffic(B)
        if live is not None:
            out["roofline"]["traffic"] = live["bytes_per_launch"]
            out["roofline"]["traffic_source"] = (
                f"measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one pass each, --kernel-trace only) over "
                f"tools/sweep_traffic_child.py ({live['launches']} launches of sk_sweep2_kernel<2, true>; FETCH_SIZE {live['fetch_size_kib']} KiB "
                f"doubled per the gfx950 note + WRITE_SIZE {live['write_size_kib']} KiB); profiles/pmc_summary.json of an earlier "
                f"profiled run says {pmc_traffic('sk_sweep_kernel')}")
            out["roofline"]["traffic_over_algorithmic"] = round(live["bytes_per_launch"] / out["roofline"]["algorithmic_bytes_per_launch"], 4)

    if use_dist:
        try:
            ops.comm_destroy()                               # barrier inside: no peer still stores into this rank's buffer
        except Exception:
            pass
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        # the LAST key of the line (what a tail of the output keeps): the second headline metric and the other legs in one
        # compact object — the full objects are `adc`, `ivf`, `index_build`, `kmeans_stats` above
        out["headline2"] = _headline2(out)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


def _valu_bound(rows, M, K, measured_ms):
    """The bound the index-build screen actually hits: vector-ALU issue.  Per candidate distance the kernel
    issues 2.25 VALU lane-instructions in its epilogue (tag 1 + pair-folded min / second-min 1.25; 36 wave-instructions per tile
    of 32 centroids x 32 documents, counted in the ISA) and 3.5 counting every VALU instruction of the tile loop (280 wave-instructions
    per 5 tiles besides the 20 MFMAs: fragment
    conversion, addressing, the running-minimum bookkeeping); the chip issues 256 CUs x 4 SIMDs x 16 lanes per clock at 2.4 GHz."""
    lane_rate = 256 * 4 * 16 * 2.4e9
    cand = rows * M * K
    floor_ms = cand * 2.25 / lane_rate * 1e3
    loop_ms = cand * 3.5 / lane_rate * 1e3
    pmc = None
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))["assign_mfma_kernel"]
        pmc = {"valu_busy_frac": sq.get("valu_busy_frac"), "mfma_busy_frac": sq.get("mfma_busy_frac"),
               "source": "profiles/pmc_summary.json (an earlier profiled run of this command), not this run"}
    except Exception:
        pass
    return {"bound": "valu-issue", "epilogue_floor_ms": round(floor_ms, 3), "frac_of_epilogue_floor": round(floor_ms / measured_ms, 3),
            "all_loop_valu_ms": round(loop_ms, 3), "frac": round(loop_ms / measured_ms, 3), "pmc": pmc,
            "what": "candidates x VALU lane-instructions per candidate / (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz): 2.25 per candidate "
                    "for the min / second-min epilogue alone, 3.5 with every VALU instruction of the tile loop; `frac` = the "
                    "latter over the measured kernel time = the share of the kernel in which the vector ALUs issue (the PMC "
                    "counter SQ_ACTIVE_INST_VALU says the same: `pmc.valu_busy_frac`)"}


def _headline2(out):
    """Compact summary (< 1000 characters) of the legs beyond the constrained assignment, placed LAST in the JSON line so that a
    2000-character tail of the output keeps it whole; the full objects are `adc`, `ivf`, `index_build`, `kmeans_stats`."""
    h = {}
    a = out.get("adc")
    if a:
        rf = a.get("roofline", {})
        h.update({"adc_queries_per_sec": a.get("value"), "adc_config": f"8841823x48B flat, {a.get('query_batch')}-query batches, k={a.get('k')}",
                  "adc_screen_ms": rf.get("avg_launch_ms"), "adc_screen_frac_lds_gather_roof": rf.get("frac"),
                  "adc_ids_equal_cpu_port": a.get("gpu_ids_identical"), "adc_score_bits_equal_cpu_port": a.get("gpu_score_bits_identical"),
                  "adc_index_built_codes_qps": (a.get("index_built_codes") or {}).get("value"),
                  "adc_cpu_port_qps": (a.get("cpu_baseline") or {}).get("value")})
    v = out.get("ivf")
    if v:
        rs = v.get("retrieval_summary") or {}
        h["ivf_m96_nlist5000_qps"] = v.get("queries_per_sec")
        h["ivf_mrr10"] = rs.get("mrr_at_10")
        h["ivf_mrr10_flat"] = rs.get("mrr_at_10_flat")
        h["ivf_nprobe_within_0.001_of_flat"] = rs.get("smallest_nprobe_with_mrr_within_0.001_of_flat")
        h["ivf_screen_frac"] = {"retrieval_index": (v.get("roofline") or {}).get("frac"),
                                "uniform_cells": ((v.get("uniform_cells") or {}).get("roofline") or {}).get("frac")}
    b = out.get("index_build")
    if b:
        h["index_build_vectors_per_sec"] = b.get("value")
        h["index_build_frac_valu_issue"] = ((b.get("roofline") or {}).get("valu_issue_bound") or {}).get("frac")
    km = out.get("kmeans_stats")
    if km:
        h["kmeans_stats_frac"] = (km.get("roofline") or {}).get("frac")
    return h


def _count_stats(st):
    """{"survivors": int32 [nq], "candidates": int32 [nq]} of ops.adc_search(stats=...) -> means / maxima for the bench line."""
    o = {}
    for key in ("survivors", "candidates"):
        if key in st:
            v = st[key].float()
            o[key] = {"mean": round(float(v.mean()), 1), "max": int(v.max()), "min": int(v.min())}
    o["what"] = ("rows per query that pass the 8-bit screen (survivors) / that the exact fp32 rescoring keeps above the sampled "
                 "threshold (candidates), one 1200-query batch")
    return o


def _faiss_or_none():
    """SURVEY 8d / BASELINE.md 4.2: the planned CPU comparator is Faiss itself when the box has it."""
    try:
        import faiss                                          # noqa: F401  (not in this image: the port runs instead)
        return faiss
    except Exception:
        return None


def _host_cpus():
    """What the host gives this process: logical CPUs, the affinity mask, a cgroup CPU quota if one is set (a quota below the
    thread count explains CPU baselines that scale badly: the OpenMP / torch threads are then time-sliced)."""
    o = {"logical_cpus": os.cpu_count()}
    try:
        o["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().strip()           # cgroup v2: "<quota us> <period us>" or "max <period>"
        o["cgroup_cpu.max"] = txt
        q, per = txt.split()
        if q != "max":
            o["quota_cpus"] = -(-int(q) // int(per))
    except Exception:
        try:                                                          # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            o["cgroup_cfs"] = f"{q} {per}"
            if q > 0:
                o["quota_cpus"] = -(-q // per)
        except Exception:
            pass
    return o


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
