"""Out-of-process probe of the native multi-rank solve (csrc/comm.hip; exchange transport from RC_COMM = ipc | rccl).

`python -m repconc_amd.dist_probe` is started by every rank of a job (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR /
MASTER_PORT from the environment, its own rendezvous port), runs the native constrained assignment and the
torch.distributed-staged one on a small batch and exits 0 iff they agree.  A caller that cannot afford a hang in an
untested communicator set-up (bench.py on a node it has never seen) runs it with a timeout and falls back to the
staged driver when it fails — the failure stays inside the child process.
"""
from __future__ import annotations

import os
import sys


def main() -> int:
    import numpy as np
    import torch
    import torch.distributed as dist

    from .sharded import TorchDistComm, assign_sinkhorn_sharded

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29550")
    share = os.environ.get("RC_BENCH_SHARE_GPU", "0") == "1"          # every rank on cuda:0 (one-GPU boxes): gloo handshake
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if share:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    if "--exchange" in sys.argv:
        # first-contact mode (tools/first_contact.py): the exchange layer alone, with NAMED failure reasons as exit codes —
        # 3 = set-up (export / IPC open / communicator) failed on some rank, 4 = a peer's data did not arrive intact,
        # 5 = an exchange timed out (RC_FLAG_COMM) — and, on success, one JSON line from rank 0 with the time of one
        # all-gather of a chain's [M/2, K] fp64 row sums (the solve's exchange unit)
        import json
        from . import _lib, ops
        os.environ["RC_COMM_STRICT"] = "1"
        try:
            kind = ops.comm_init(transport=os.environ.get("RC_COMM", "ipc"))
        except Exception as e:                                  # collective: every rank lands here together
            if rank == 0:
                print(json.dumps({"ok": False, "reason": f"set-up failed: {e}"}), flush=True)
            dist.destroy_process_group()
            return 3
        code = 0
        for n in (24 * 256, 7, 1 << 18):                      # a chain's row sums, an odd size, several slots
            mine = (torch.arange(n, device=dev, dtype=torch.float64) * 1e-3 + (rank + 1) * 1000.0)
            got = ops.comm_allgather(mine)
            want = torch.stack([torch.arange(n, device=dev, dtype=torch.float64) * 1e-3 + (r + 1) * 1000.0 for r in range(world)])
            if not torch.equal(got, want):
                code = 4
        rows_ = torch.zeros((24, 256), dtype=torch.float64, device=dev)
        for _ in range(20):
            ops.comm_allgather(rows_)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.comm_allgather(rows_)
        e1.record()
        torch.cuda.synchronize()
        try:
            ops.comm_check()
        except _lib.RepconcHipError:
            code = 5
        flag = torch.tensor([code], dtype=torch.int32, device=torch.device("cpu") if share else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        code = int(flag.item())
        if rank == 0:
            print(json.dumps({"ok": code == 0, "transport": kind, "alloc": os.environ.get("RC_IPC_ALLOC", "finegrained"),
                              "world": world, "devices": 1 if share else world,
                              "us_per_allgather": round(e0.elapsed_time(e1) * 1e3 / 200, 2),
                              "reason": {0: None, 4: "a peer's data did not arrive intact (store visibility)",
                                         5: "an exchange timed out (RC_FLAG_COMM)"}[code]}), flush=True)
        ops.comm_destroy()
        dist.destroy_process_group()
        return code
    rng = np.random.default_rng(4242)
    M, K, D, rows = 48, 256, 768, 512
    x = rng.standard_normal((rows * world, D), dtype=np.float32)
    C = torch.from_numpy(np.ascontiguousarray(x[:K].reshape(K, M, D // M).transpose(1, 0, 2))).to(dev)
    xl = torch.from_numpy(x[rank * rows:(rank + 1) * rows]).to(dev)
    comm = TorchDistComm()
    os.environ["RC_DIST_NATIVE"] = "0"
    staged, _ = assign_sinkhorn_sharded(xl, C, 0.003, 100, comm, dtype=torch.uint8)
    os.environ["RC_DIST_NATIVE"] = "1"
    native, _ = assign_sinkhorn_sharded(xl, C, 0.003, 100, comm, dtype=torch.uint8)       # captures the iteration graph
    replay, _ = assign_sinkhorn_sharded(xl, C, 0.003, 100, comm, dtype=torch.uint8)       # replays it (same workspace block)
    torch.cuda.synchronize()
    ok = torch.tensor([int(torch.equal(staged, native) and torch.equal(staged, replay))], dtype=torch.int32,
                      device=torch.device("cpu") if share else dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    from . import ops
    ops.comm_destroy()
    dist.destroy_process_group()
    return 0 if int(ok.item()) == 1 else 1


if __name__ == "__main__":
    sys.exit(main())
