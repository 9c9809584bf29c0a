"""Time rc_kmeans_stats (sufficient statistics of a Lloyd iteration) at the warm-up's and a corpus chunk's size.
    python tools/kmeans_bench.py [M ...]      # default 48
Prints ms per call (HIP events around 10 calls) and the fraction of the 8 TB/s HBM roof for 4 D + M bytes per vector."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import ops  # noqa: E402

D = 768
Ms = [int(a) for a in sys.argv[1:]] or [48]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(20228)
for M in Ms:
    for rows in (1 << 16, 1 << 20, 8841823 // 8):
        x = torch.randn((rows, D), device=dev, generator=g)
        c = torch.randint(0, 256, (rows, M), dtype=torch.uint8, device=dev, generator=g)
        for _ in range(3):
            ops.kmeans_stats(x, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.kmeans_stats(x, c)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = rows * (D * 4 + M) / 1e9
        print(f"kmeans_stats M={M} rows={rows}: {ms:.4f} ms  {rows / ms / 1e3:.1f} M vec/s  {gb / ms:.2f} TB/s  frac {gb / ms / 8.0:.3f}", flush=True)
        del x, c
