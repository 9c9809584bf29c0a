"""How often a query is repeated at a given head-room of the sampled threshold (development tool, GPU; round 6): 288 000 fresh
queries per setting, M = 48 and 32, k = 1000 — against the Poisson model of ops.ADC_SEL_SLACK (3: 8e-6 per query, 2: 1.1e-4)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd.index import PQIndex
dev = "cuda:0"
N, D, QB = 8841823, 768, 1200
for M in (48, 32):
    g = torch.Generator(device=dev).manual_seed(777 + M)
    idx = PQIndex(D, M)
    idx.set_centroids(torch.randn(M, 256, D // M, device=dev, generator=g))
    idx.add_codes(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g))
    for slack in (3.0, 2.0):
        idx.sel_slack = slack
        rep = tot = 0
        t0 = time.perf_counter()
        for round_ in range(10):
            q = torch.randn(24 * QB, D, device=dev, generator=g)
            pend = []
            for b in range(24):
                fin = idx.search_async(q[b * QB:(b + 1) * QB], 1000)
                pend.append((fin, idx.last_search))
            for fin, p in pend:
                fin(); rep += p.stats["retried_queries"]
            tot += 24 * QB
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"M={M} sel_slack={slack}: {rep} of {tot} queries repeated ({rep/tot:.2e} per query); {tot/dt/1e3:.1f} k QPS incl. query generation", flush=True)
    del idx
