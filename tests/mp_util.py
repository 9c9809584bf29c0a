"""Launch `world` rank PROCESSES of tests/ipc_rank_worker.py (one per GPU when enough are visible, else all of them on
cuda:0 — the IPC transport of csrc/comm.hip works between processes that share a device) with a hard deadline: a rank
that hangs or dies takes the others down by their exact PIDs, never a device queue."""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "ipc_rank_worker.py")


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(world: int, args, outdir: str, timeout: float = 300.0, env_extra=None):
    """Run the worker with `args` on ranks 0 .. world-1; returns the list of exit codes (raises on a timeout)."""
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RC_TEST_OUTDIR=str(outdir), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.update(env_extra or {})
        log = open(os.path.join(outdir, f"rank{rank}.log"), "w")
        procs.append((subprocess.Popen([sys.executable, WORKER] + [str(a) for a in args], cwd=ROOT, env=env, stdout=log,
                                       stderr=subprocess.STDOUT), log))
    deadline = time.time() + timeout
    codes = [None] * world
    try:
        while any(c is None for c in codes):
            for i, (p, _) in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if any(c not in (None, 0) for c in codes):
                break                                   # a rank failed: the others would wait for it until their own timeout
            if time.time() > deadline:
                raise TimeoutError(f"ranks still running after {timeout} s: {[i for i, c in enumerate(codes) if c is None]}")
            time.sleep(0.05)
    finally:
        for p, log in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
            log.close()
    return [p.returncode for p, _ in procs]


def rank_logs(outdir: str, world: int) -> str:
    out = []
    for r in range(world):
        try:
            out.append(f"--- rank {r}\n" + open(os.path.join(outdir, f"rank{r}.log")).read()[-3000:])
        except OSError:
            pass
    return "\n".join(out)
