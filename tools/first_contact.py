#!/usr/bin/env python3
"""First contact with a multi-GPU node: which exchange transport of csrc/comm.hip works here, and which is faster.

    python tools/first_contact.py [world]          # default: one rank per visible GPU (>= 2), else 2 ranks sharing cuda:0

For every candidate — the IPC transport with its receive buffer in fine-grained, uncached and plain device memory
(RC_IPC_ALLOC), then RCCL — `world` rank processes run `python -m repconc_amd.dist_probe --exchange`: set-up (cross-device
hipIpcOpenMemHandle of the buffer / communicator creation), three all-gathers whose every byte is checked (peer stores
visible under the acquire load of the arrival counter), 200 timed all-gathers of a Sinkhorn chain's row sums.  A candidate
that hangs is killed after 120 s (its ranks only; exact PIDs).  Prints one line per candidate with the NAMED reason of a
failure, then the recommendation (what RC_COMM=auto will pick is ipc/finegrained, else rccl; an uncached / plain winner is
an RC_IPC_ALLOC setting to make the default)."""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(world, share, env_extra, timeout=120.0):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RC_IPC_TIMEOUT_MS="8000", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        if share:
            env["RC_BENCH_SHARE_GPU"] = "1"
        env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable, "-m", "repconc_amd.dist_probe", "--exchange"], cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=subprocess.DEVNULL, text=True))
    deadline = time.time() + timeout
    while any(p.poll() is None for p in procs) and time.time() < deadline:
        time.sleep(0.1)
    hung = [i for i, p in enumerate(procs) if p.poll() is None]
    for p in procs:
        if p.poll() is None:
            p.kill()
            p.wait()
    out = procs[0].stdout.read() if procs[0].stdout else ""
    line = next((ln for ln in out.splitlines() if ln.startswith("{")), None)
    res = json.loads(line) if line else {"ok": False, "reason": None}
    if hung:
        res = {"ok": False, "reason": f"hung: ranks {hung} killed after {timeout:.0f} s"}
    elif not res.get("ok") and not res.get("reason"):
        res["reason"] = f"rank exit codes {[p.returncode for p in procs]}"
    return res


def main():
    import torch
    ndev = torch.cuda.device_count()
    world = int(sys.argv[1]) if len(sys.argv) > 1 else (ndev if ndev >= 2 else 2)
    share = ndev < world
    print(f"{ndev} device(s) visible; {world} ranks, " + ("all on cuda:0 (shared-GPU walk: same code path, no xGMI)" if share else "one per GPU"))
    cands = [("ipc", "finegrained"), ("ipc", "uncached"), ("ipc", "plain")] + ([] if share else [("rccl", "-")])
    results = {}
    for transport, alloc in cands:
        env = {"RC_COMM": transport}
        if transport == "ipc" and alloc != "finegrained":
            env["RC_IPC_ALLOC"] = alloc
        r = run(world, share, env)
        results[(transport, alloc)] = r
        what = f"{r['us_per_allgather']:8.2f} us per all-gather" if r.get("ok") else f"FAILED: {r.get('reason')}"
        print(f"  {transport:5s} {alloc:12s} {what}", flush=True)
    ok = {k: v for k, v in results.items() if v.get("ok")}
    if not ok:
        print("no native transport works here: the solve falls back to the python-staged torch.distributed loop (RC_DIST_NATIVE=0)")
        return 1
    best = min(ok, key=lambda k: ok[k]["us_per_allgather"])
    print(f"fastest: {best[0]} / {best[1]} ({ok[best]['us_per_allgather']} us); RC_COMM=auto picks " +
          ("ipc / finegrained" if ("ipc", "finegrained") in ok else "rccl" if ("rccl", "-") in ok else "nothing native"))
    print(json.dumps({f"{k[0]}/{k[1]}": v for k, v in results.items()}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
