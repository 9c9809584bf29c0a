#!/usr/bin/env python3
"""OPQ + PQ warm-up timing (development tool, GPU): 65 536 training rows x 768, M = 48 (a-12)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import ops  # noqa: E402
from repconc_amd.train.run_warmup import train_opq, train_pq  # noqa: E402

dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 48
g = torch.Generator(device=dev).manual_seed(3)
centres = torch.randn(2048, 768, device=dev, generator=g)
x = centres[torch.randint(0, 2048, (65536,), device=dev, generator=g)] + 0.7 * torch.randn(65536, 768, device=dev, generator=g)
torch.cuda.synchronize()
t0 = time.perf_counter()
C, mse = train_pq(x, M, 25)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"train_pq 25 Lloyd iterations on 65536 x 768, M={M}: {(t1 - t0) * 1e3:.1f} ms, mse {mse:.4f}")
for name, fn in (("assign_nearest", lambda: ops.assign_nearest(x, C, torch.uint8)),
                 ("kmeans_stats", lambda: ops.kmeans_stats(x, codes))):
    codes = ops.assign_nearest(x, C, torch.uint8)
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"  {name}: {(time.perf_counter() - t) * 100:.3f} ms per call")
t0 = time.perf_counter()
R = train_opq(x, M, n_outer=50)
torch.cuda.synchronize()
print(f"train_opq 50 outer iterations: {(time.perf_counter() - t0):.2f} s")
