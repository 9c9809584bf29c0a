import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import ops
rng = np.random.default_rng(123)
dev = "cuda:0"
bad = 0
for trial in range(120):
    M = int(rng.choice([8, 12, 16, 24, 32, 48, 64, 96]))
    B = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 300, 1000, 4097, 20000]))
    scale = float(rng.choice([1e-3, 1.0, 50.0, 1e4]))
    kind = rng.integers(0, 4)
    x = rng.standard_normal((B, 768), dtype=np.float32) * scale
    C = rng.standard_normal((M, 256, 768 // M), dtype=np.float32) * scale
    if kind == 1:   # centroids sampled from data (exact zeros distances)
        take = rng.integers(0, B, 256)
        C = np.ascontiguousarray(x[take].reshape(256, M, 768 // M).transpose(1, 0, 2))
    if kind == 2:   # duplicated centroids
        C[:, 128:] = C[:, :128]
    if kind == 3:   # zeros
        x[: B // 2] = 0
        C[:, :7] = 0
    xt, Ct = torch.from_numpy(x).to(dev), torch.from_numpy(C).to(dev)
    a = ops.assign_nearest(xt, Ct, torch.uint8, method="exact")
    b = ops.assign_nearest(xt, Ct, torch.uint8, method="mfma")
    if not torch.equal(a, b):
        bad += 1
        print("MISMATCH", M, B, scale, kind, int((a != b).sum()))
print("trials done, mismatching shapes:", bad)
