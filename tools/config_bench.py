#!/usr/bin/env python3
"""Timing of the whole constrained / nearest assignment for the BASELINE.json config shapes (development tool)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import ops  # noqa: E402

dev = "cuda:0"
for name, B, M in (("c1 10k x768 M=8", 10000, 8), ("c2/c3 49152 M=48", 49152, 48), ("c3 per-rank 6144 M=48", 6144, 48),
                   ("c4 49152 M=96", 49152, 96), ("c5 49152 M=24", 49152, 24), ("M=64 49152", 49152, 64)):
    x = torch.randn(B, 768, device=dev)
    C = x[torch.randperm(B, device=dev)[:256]].reshape(256, M, 768 // M).transpose(0, 1).contiguous()
    for _ in range(2):
        codes, fl = ops.assign_sinkhorn(x, C, 0.003, 100, torch.uint8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        codes, fl = ops.assign_sinkhorn(x, C, 0.003, 100, torch.uint8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    h = ops.code_hist(codes).float()
    imb = float((h / (B / 256) - 1).abs().max())
    t0 = time.perf_counter()
    for _ in range(n):
        near = ops.assign_nearest(x, C, torch.uint8)
    torch.cuda.synchronize()
    dn = (time.perf_counter() - t0) / n
    print(f"{name:26s} constrained {dt*1e3:8.2f} ms ({B/dt/1e3:8.1f} k vec/s, {B*M/dt/1e6:7.2f} M sub/s, max imbalance {imb:.3f}, "
          f"flags {int(fl.item())}) | nearest {dn*1e3:7.3f} ms ({B/dn/1e6:6.2f} M vec/s)")
