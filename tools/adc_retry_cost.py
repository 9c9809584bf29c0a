"""What a repeated query of the flat search costs (development tool, GPU; round 6): sel_slack 2 provokes repeats; per batch the
enqueue / result times of the batches that repeated a query, then the same repeat in isolation."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd.index import PQIndex
from repconc_amd import ops
dev = "cuda:0"
N, D, M, QB, NB = 8841823, 768, 48, 1200, 24
g = torch.Generator(device=dev).manual_seed(20222)
idx = PQIndex(D, M)
idx.set_centroids(torch.randn(M, 256, D // M, device=dev, generator=g))
idx.add_codes(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g))
q = torch.randn(NB * QB, D, device=dev, generator=g)
idx.sel_slack = 2.0
idx.search(q[:QB], 1000); torch.cuda.synchronize()
# one batch at a time, synchronously: which batches retry and what a retry costs
for b in range(NB):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fin = idx.search_async(q[b * QB:(b + 1) * QB], 1000); p = idx.last_search
    t1 = time.perf_counter()
    fin(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    if p.stats["retried_queries"] or b < 2:
        print(f"batch {b}: enqueue {(t1-t0)*1e3:.2f} ms, result {(t2-t1)*1e3:.2f} ms, retried {p.stats['retried_queries']} exact {p.stats['exact_queries']}", flush=True)
# the same retry in isolation
import torch.autograd.profiler as prof
bq = None
for b in range(NB):
    fin = idx.search_async(q[b * QB:(b + 1) * QB], 1000); p = idx.last_search
    fin()
    if p.stats["retried_queries"]:
        bq = b; break
if bq is not None:
    qq = q[bq * QB:(bq + 1) * QB]
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pend = ops.adc_search(idx.codes, idx._centroids, qq, 1000, scan_image=idx._image, defer=True, sel_slack=2.0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        st = int(pend._status.item()); bad = torch.nonzero(pend._qstatus).flatten()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        s, i, qs = pend._rerun(bad, 8.0, False)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f"rep {rep}: first pass {(t1-t0)*1e3:.2f} ms, status read {(t2-t1)*1e3:.2f} ms, rerun of {bad.numel()} queries at slack 8: {(t3-t2)*1e3:.2f} ms, still bad {int((qs!=0).sum())}", flush=True)
