#!/usr/bin/env python3
"""IVF: head-room of the sampled threshold (development tool, GPU; round 6).  M = 96, 5000 cells, 6 980-query calls, nprobe 8 / 32 /
128, SEL_SLACK in argv (default 6 5 4 3): ms per search and queries answered again by the per-query scan."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd.ivf import IVFPQIndex  # noqa: E402

dev = "cuda:0"
N, M, nlist, k, nq = 8841823, 96, 5000, 1000, 6980
slacks = [float(a) for a in sys.argv[1:]] or [6.0, 5.0, 4.0, 3.0]
g = torch.Generator(device=dev).manual_seed(1)
ivf = IVFPQIndex(768, M, nlist, device=dev)
ivf.set_centroids(torch.randn(M, 256, 768 // M, device=dev, generator=g))
ivf.coarse = torch.randn(nlist, 768, device=dev, generator=g)
ivf.set_lists(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g), torch.randint(0, nlist, (N,), device=dev, generator=g))
qs = [torch.randn(nq, 768, device=dev, generator=g) for _ in range(6)]
calls = {"n": 0}
orig_search = ivf.search


def counting(x, kk, nprobe=None, method="auto"):
    if method == "scan":
        calls["n"] += int(x.shape[0])
    return orig_search(x, kk, nprobe, method)
ivf.search = counting
for nprobe in (8, 32, 128):
    for slack in slacks:
        ivf.SEL_SLACK = slack
        ivf.search(qs[0], k, nprobe)
        torch.cuda.synchronize()
        calls["n"] = 0
        t0 = time.perf_counter()
        for q in qs:
            ivf.search(q, k, nprobe)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(qs)
        print(f"nprobe={nprobe:3d} SEL_SLACK={slack:3.1f}: {dt*1e3:7.2f} ms per {nq}-query search = {nq/dt/1e3:7.1f} k QPS; "
              f"queries answered again by the scan: {calls['n']} of {nq*len(qs)}", flush=True)
