#!/usr/bin/env python3
"""Head-room of the sampled candidate threshold (development tool, GPU; round 6): queries/s, survivors of the 8-bit screen,
candidates after rescoring and REPEATED queries for sel_slack in argv (default 6 4 3 2 1), flat M = 48, 8.84 M rows, k = 1000,
24 batches of 1200 fresh queries each (a repeated query costs a search of its own)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd.index import PQIndex  # noqa: E402

dev = "cuda:0"
N, D, M, QB, NB = 8841823, 768, 48, 1200, 24
slacks = [float(a) for a in sys.argv[1:]] or [6.0, 4.0, 3.0, 2.0, 1.0]
g = torch.Generator(device=dev).manual_seed(20222)
idx = PQIndex(D, M)
idx.set_centroids(torch.randn(M, 256, D // M, device=dev, generator=g))
idx.add_codes(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g))
q = torch.randn(NB * QB, D, device=dev, generator=g)
for k in (1000, 10):
    for slack in slacks:
        idx.sel_slack = slack
        idx.search(q[:QB], k)
        torch.cuda.synchronize()
        retried = exact = 0
        t0 = time.perf_counter()
        pend = []
        for b in range(NB):
            fin = idx.search_async(q[b * QB:(b + 1) * QB], k)
            pend.append((fin, idx.last_search))
        for fin, p in pend:
            fin()
            retried += p.stats["retried_queries"]
            exact += p.stats["exact_queries"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = {}
        idx.search_async(q[:QB], k, stats=st)()
        print(f"k={k:4d} sel_slack={slack:3.1f}: {NB*QB/dt/1e3:7.1f} k QPS, {dt/NB*1e3:6.2f} ms per batch; survivors per query "
              f"{float(st['survivors'].float().mean()):7.0f}, candidates {float(st['candidates'].float().mean()):7.0f}; "
              f"repeated queries {retried} of {NB*QB}, exact-path queries {exact}", flush=True)
