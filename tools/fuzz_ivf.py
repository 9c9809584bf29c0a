import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd.ivf import IVFPQIndex
from repconc_amd.index import PQIndex
# python tools/fuzz_ivf.py [seed [trials]] — round 6: every trial also runs BOTH widths of the list-centric screen (lists8 / lists16)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 321)
dev = "cuda:0"
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    M = int(rng.choice([16, 32, 48, 64, 96]))
    N = int(rng.choice([3000, 50000, 270000, 600000]))
    nlist = int(rng.choice([7, 64, 500, 3000]))
    nq = int(rng.choice([1, 5, 17, 33, 200, 700]))
    k = int(rng.choice([1, 10, 100, 1000]))
    nprobe = int(min(nlist, rng.choice([1, 3, 16, 64, 10000])))
    codes = torch.from_numpy(rng.integers(0, 256, (N, M), dtype=np.uint8)).to(dev)
    if rng.integers(0, 2):
        cells = np.minimum((rng.pareto(1.1, N) * 2).astype(np.int64), nlist - 1)
    else:
        cells = rng.integers(0, nlist, N)
    ivf = IVFPQIndex(768, M, nlist, device=dev)
    ivf.set_centroids(torch.from_numpy(rng.standard_normal((M, 256, 768 // M), dtype=np.float32)))
    ivf.coarse = torch.from_numpy(rng.standard_normal((nlist, 768), dtype=np.float32)).to(dev)
    ivf.set_lists(codes, torch.from_numpy(cells).to(dev))
    q = torch.from_numpy(rng.standard_normal((nq, 768), dtype=np.float32)).to(dev)
    s1, i1 = ivf.search(q, k, nprobe, method="lists")
    s2, i2 = ivf.search(q, k, nprobe, method="scan")
    ok = torch.equal(i1, i2) and torch.equal(s1, s2)
    for method in ("lists8", "lists16"):
        sw, iw = ivf.search(q, k, nprobe, method=method)
        ok = ok and torch.equal(iw, i2) and torch.equal(sw, s2)
    if nprobe == nlist and N <= 270000:
        flat = PQIndex(768, M, device=dev); flat.set_centroids(ivf.pq_centroids); flat.add_codes(codes)
        fs, fi = flat.search(q, k)
        ok = ok and torch.equal(fs, s1)
    if not ok:
        bad += 1
        print("MISMATCH", M, N, nlist, nq, k, nprobe)
print("ivf trials done, mismatches:", bad)
