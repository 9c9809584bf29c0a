#!/usr/bin/env python3
"""SURVEY 8d-D sweep (development tool, GPU): ADC search over the 8.84 M-row index for M in {48, 96}, k in {10, 200, 1000},
6980 queries in batches of 1200, with i.i.d. uniform codes and with codes produced by the index-build path (8d-C:
nearest-code assignment of clustered synthetic vectors, so the code distribution is skewed like a real index)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import ops  # noqa: E402

dev = "cuda:0"
N, D, NQ, QB = 8841823, 768, 6980, 1200
g = torch.Generator(device=dev).manual_seed(20222)
q = torch.randn(NQ, D, device=dev, generator=g)
for M in (48, 96):
    C = torch.randn(M, 256, D // M, device=dev, generator=g)
    uni = torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g)
    # 8d-C: codes of a clustered corpus (mixture of 4096 centres + noise), built in 1 M-row chunks
    centres = torch.randn(4096, D, device=dev, generator=g)
    built = torch.empty((N, M), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tb = 0.0
    for s in range(0, N, 1 << 20):
        n = min(1 << 20, N - s)
        x = centres[torch.randint(0, 4096, (n,), device=dev, generator=g)] + 0.5 * torch.randn(n, D, device=dev, generator=g)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        built[s:s + n] = ops.assign_nearest(x, C, torch.uint8)
        torch.cuda.synchronize()
        tb += time.perf_counter() - t1
    print(f"M={M}: index build (nearest codes of {N} x {D}) {tb*1e3:.1f} ms = {N/tb/1e6:.1f} M vectors/s", flush=True)
    for label, codes in (("uniform codes", uni), ("built codes", built)):
        for k in (10, 200, 1000):
            ops.adc_search(codes, C, q[:QB], k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(0, NQ, QB):
                sc, ids = ops.adc_search(codes, C, q[s:s + QB], k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"M={M} {label:13s} k={k:4d}: {NQ} queries in {dt*1e3:8.1f} ms = {NQ/dt/1e3:6.1f} k QPS", flush=True)
    del uni, built
