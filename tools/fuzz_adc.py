import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import ops
from repconc_amd.index import PQIndex
from oracle import c_oracle
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 777)
dev = "cuda:0"
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    M = int(rng.choice([8, 16, 24, 32, 48, 64, 96]))
    N = int(rng.choice([262144, 300001, 327680, 327681, 700000, 1200000, 2049 * 1024 + 5]))
    nq = int(rng.choice([1, 7, 15, 16, 17, 33, 40, 130]))
    k = int(rng.choice([1, 10, 200, 1000, 3000]))
    codes = rng.integers(0, 256, (N, M), dtype=np.uint8)
    if trial % 3 == 0:
        codes[: N // 2] = codes[N // 2: N // 2 * 2]       # duplicates: ties
    C = rng.standard_normal((M, 256, 768 // M), dtype=np.float32)
    q = rng.standard_normal((nq, 768), dtype=np.float32)
    idx = PQIndex(768, M, device=dev); idx.set_centroids(torch.from_numpy(C))
    cut = int(rng.integers(1, N))                           # appended in two pieces: the image is extended in place
    idx.add_codes(torch.from_numpy(codes[:cut]).to(dev)); idx.add_codes(torch.from_numpy(codes[cut:]).to(dev))
    s, i = idx.search(torch.from_numpy(q).to(dev), k)
    ws, wi = c_oracle.adc_search(codes, C, q, k)
    ok = np.array_equal(i.cpu().numpy(), wi) and np.array_equal(s.cpu().numpy().view(np.uint32), ws.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH", M, N, nq, k)
print("adc trials done, mismatches:", bad)
