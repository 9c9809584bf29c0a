#!/usr/bin/env python3
"""IVF (nlist = 5000) ADC search timing at the BASELINE index size (development tool; SURVEY 8d input D).
Synthetic: uniform random codes, rows assigned to cells uniformly at random, Gaussian coarse centroids/queries."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd.index import PQIndex  # noqa: E402
from repconc_amd.ivf import IVFPQIndex  # noqa: E402

dev = "cuda:0"
N, M, nlist, k = 8841823, int(sys.argv[1]) if len(sys.argv) > 1 else 48, 5000, 1000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1200       # 6980 = the MS MARCO dev set in one call (batch_search does that for IVF)
g = torch.Generator(device=dev).manual_seed(1)
codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g)
cells = torch.randint(0, nlist, (N,), device=dev, generator=g)
C = torch.randn(M, 256, 768 // M, device=dev, generator=g)
ivf = IVFPQIndex(768, M, nlist, device=dev)
ivf.set_centroids(C)
ivf.coarse = torch.randn(nlist, 768, device=dev, generator=g)
ivf.set_lists(codes, cells)
q = torch.randn(nq, 768, device=dev, generator=g)
flat = PQIndex(768, M, device=dev)
flat.set_centroids(C)
flat.add_codes(codes)
fs, fi = flat.search(q, k)
torch.cuda.synchronize()
t0 = time.perf_counter(); fs, fi = flat.search(q, k); torch.cuda.synchronize(); tf = time.perf_counter() - t0
print(f"flat: {tf*1e3:.1f} ms per {nq} queries = {nq/tf:.0f} QPS")
for nprobe in (1, 8, 32, 128, 512):
    for _ in range(2):
        s, i = ivf.search(q, k, nprobe)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        s, i = ivf.search(q, k, nprobe)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    rec = (i[:, :10].unsqueeze(2) == fi[:, :10].unsqueeze(1)).any(2).float().mean().item()
    print(f"nprobe={nprobe:4d}: {dt*1e3:8.1f} ms per {nq} queries = {nq/dt:9.0f} QPS, rows scanned/query ~{N*nprobe//nlist}, "
          f"top-10 overlap with flat {rec:.3f} (random cells: expected ~nprobe/nlist)")
