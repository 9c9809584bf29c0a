"""IVF: size of the per-query sample that places the candidate threshold (development tool, GPU; round 6): ms per search for
SAMPLE_ROWS in (12288, 6144, 3072, 1536), 6 980- and 1 200-query calls, nprobe 8 / 32 / 128."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd.ivf import IVFPQIndex
dev = "cuda:0"
N, M, nlist, k, nq = 8841823, 96, 5000, 1000, 6980
g = torch.Generator(device=dev).manual_seed(1)
ivf = IVFPQIndex(768, M, nlist, device=dev)
ivf.set_centroids(torch.randn(M, 256, 768 // M, device=dev, generator=g))
ivf.coarse = torch.randn(nlist, 768, device=dev, generator=g)
ivf.set_lists(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g), torch.randint(0, nlist, (N,), device=dev, generator=g))
qs = [torch.randn(nq, 768, device=dev, generator=g) for _ in range(6)]
calls = {"n": 0}
orig = ivf.search
def counting(x, kk, nprobe=None, method="auto"):
    if method == "scan":
        calls["n"] += int(x.shape[0])
    return orig(x, kk, nprobe, method)
ivf.search = counting
for nqq in (6980, 1200):
    for nprobe in (8, 32, 128):
        for sr in (12288, 6144, 3072, 1536):
            ivf.SAMPLE_ROWS = sr
            ivf.search(qs[0][:nqq], k, nprobe); torch.cuda.synchronize()
            calls["n"] = 0
            t0 = time.perf_counter()
            for q in qs:
                ivf.search(q[:nqq], k, nprobe)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / len(qs)
            print(f"nq={nqq} nprobe={nprobe:3d} SAMPLE_ROWS={sr:5d} (ss={ivf._sample_step(nprobe)}): {dt*1e3:7.2f} ms = {nqq/dt/1e3:7.1f} k QPS; scan-answered {calls['n']}", flush=True)
