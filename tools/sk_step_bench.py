#!/usr/bin/env python3
"""Per-step time of the constrained assignment at a few shapes, many repetitions (development tool, GPU):
    python tools/sk_step_bench.py [reps]      A/B libraries via REPCONC_HIP_LIB."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import ops  # noqa: E402
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for B, M in ((6144, 48), (6144, 24), (1024, 48), (12288, 48), (49152, 48)):
    x = torch.randn(B, 768, device=dev)
    C = x[torch.randperm(B, device=dev)[:256]].reshape(256, M, 768 // M).transpose(0, 1).contiguous()
    for _ in range(3):
        codes, fl = ops.assign_sinkhorn(x, C, 0.003, 100, torch.uint8)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            codes, fl = ops.assign_sinkhorn(x, C, 0.003, 100, torch.uint8)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    print(f"{B:6d} x {M:2d}: {best*1e3:7.3f} ms per step, crc {int(codes.sum().item())}", flush=True)
