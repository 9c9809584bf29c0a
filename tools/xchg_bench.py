#!/usr/bin/env python3
"""Per-rank shape of the 8-GPU recipe (6 144 x 768, M = 48) on ONE GPU with the exchange of every iteration forced
(RC_DIST_FORCE_COLL=1 on a one-rank IPC transport: the rank pushes to and waits for itself) — the proxy of DESIGN.md §5 /
VERDICT r4 item 1.  ms per step (100 iterations), best of 3 x 20 steps, for every form of the exchange, next to the
stand-alone solve without any exchange.

    python tools/xchg_bench.py [rows [M]]
"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from repconc_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
M = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(B, 768, device=dev, generator=g)
C = x[:256].reshape(256, M, 768 // M).transpose(0, 1).contiguous()
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
assert ops.comm_init(transport="ipc") == "ipc"
KEYS = ("RC_DIST_FORCE_COLL", "RC_IPC_XSWEEP", "RC_IPC_INWAIT", "RC_DIST_SPLIT")
SOLO = "solo"
forms = [
    ("stand-alone (rc_pq_assign_sinkhorn), no exchange", {SOLO: "1"}),
    ("stand-alone, two chains (RC_DIST_SPLIT=1)", {SOLO: "1", "RC_DIST_SPLIT": "1"}),
    ("one chain, fused exchange, wait in the prologue", {"RC_DIST_FORCE_COLL": "1", "RC_IPC_INWAIT": "1", "RC_DIST_SPLIT": "0"}),
    ("one chain, fused exchange, flag-wait kernel", {"RC_DIST_FORCE_COLL": "1", "RC_IPC_INWAIT": "0", "RC_DIST_SPLIT": "0"}),
    ("two chains, fused exchange (round 6: two chains always use the flag-wait kernels)", {"RC_DIST_FORCE_COLL": "1", "RC_IPC_INWAIT": "1", "RC_DIST_SPLIT": "1"}),
    ("one chain, push + wait kernel (rounds 3-4)", {"RC_DIST_FORCE_COLL": "1", "RC_IPC_XSWEEP": "0", "RC_DIST_SPLIT": "0"}),
    ("two chains, push + wait kernels (rounds 3-4)", {"RC_DIST_FORCE_COLL": "1", "RC_IPC_XSWEEP": "0", "RC_DIST_SPLIT": "1"}),
]
res, ref = {}, None
for rep in range(3):
    for name, env in forms:
        for k in KEYS:
            os.environ.pop(k, None)
        solo = SOLO in env
        os.environ.update({k: v for k, v in env.items() if k != SOLO})
        fn = (lambda: ops.assign_sinkhorn(x, C, 0.003, 100, torch.uint8)) if solo else \
             (lambda: ops.assign_sinkhorn_dist(x, C, 0.003, 100, torch.uint8))
        for _ in range(3):
            codes, fl = fn()
        torch.cuda.synchronize()
        assert int(fl.item()) == 0
        ref = codes.clone() if ref is None else ref
        if not torch.equal(codes, ref):
            if os.environ.get("XB_NOCHECK"):
                print("  (codes differ in:", name, ")")
            else:
                raise AssertionError(name)
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        res.setdefault(name, []).append((time.perf_counter() - t0) / 20 * 1e3)
print(f"{B} x 768, M = {M}, eps 0.003, T = 100; ms per step (best of 3 x 20), codes identical in every form")
for k, v in res.items():
    print(f"  {k:56s} {min(v):.3f}", flush=True)
ops.comm_destroy(group_barrier=False)
dist.destroy_process_group()
