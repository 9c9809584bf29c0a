#!/usr/bin/env python3
"""Round-5 measurement of the RESIDENT form of the sweeps (experimental build, not shipped): sweeps 2 .. T-1 of
one rank in ONE launch whose blocks stay on the chip and synchronise per sub-quantiser (tools/_exp/resident5, the round-4 kernel
ported onto the round-5 sources).  Run with REPCONC_HIP_LIB pointing at the variant library:
    REPCONC_HIP_LIB=build/var/resident.so python tools/resident_bench.py [rows ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from repconc_amd import ops  # noqa: E402

dev = "cuda:0"
sizes = [int(a) for a in sys.argv[1:]] or [6144, 49152]
for B in sizes:
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, 768, device=dev, generator=g)
    C = x[:256].reshape(256, 48, 16).transpose(0, 1).contiguous()
    ref = None
    for name, env in (("one launch per sweep (shipped)", {"RC_SK_RESIDENT": "0"}),
                      ("resident, a block alternates between two sub-quantisers, clock-rotating priority", {"RC_SK_RESIDENT": "1", "RC_SK_PRIO": "20"}),
                      ("resident, one sub-quantiser per block", {"RC_SK_RESIDENT": "1", "RC_SK_PRIO": "20", "RC_SK_RES_PAIR": "0"}),
                      ("resident, two sub-quantisers per block, no priority", {"RC_SK_RESIDENT": "1", "RC_SK_PRIO": "0"})):
        for k in ("RC_SK_RESIDENT", "RC_SK_PRIO", "RC_SK_RES_PAIR"):
            os.environ.pop(k, None)
        os.environ.update(env)
        best = 1e9
        for rep in range(3):
            for _ in range(3):
                codes, fl = ops.assign_sinkhorn(x, C, 0.003, 100, torch.uint8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20 if B <= 8192 else 8):
                codes, fl = ops.assign_sinkhorn(x, C, 0.003, 100, torch.uint8)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / (20 if B <= 8192 else 8) * 1e3)
        ref = codes.clone() if ref is None else ref
        same = int((codes != ref).sum().item())
        print(f"{B} x 768, M = 48: {name:86s} {best:8.3f} ms per step   flags {int(fl.item())}   codes differing from the shipped path: {same}", flush=True)
