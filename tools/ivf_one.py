"""One IVF configuration (8.84 M x 96 B, 5000 cells, argv[2] = 1200 queries, k = 1000) searched 20 times at nprobe = argv[1] (default 128):
the command the PMC passes of the IVF screen are collected over (tools/pmc_collect.sh)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd.ivf import IVFPQIndex
dev = "cuda:0"
N, M, nlist, nq, k = 8841823, 96, 5000, 1200, 1000
nprobe = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nq = int(sys.argv[2]) if len(sys.argv) > 2 else nq
g = torch.Generator(device=dev).manual_seed(1)
codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g)
cells = torch.randint(0, nlist, (N,), device=dev, generator=g)
ivf = IVFPQIndex(768, M, nlist, device=dev)
ivf.set_centroids(torch.randn(M, 256, 768 // M, device=dev, generator=g))
ivf.coarse = torch.randn(nlist, 768, device=dev, generator=g)
ivf.set_lists(codes, cells)
q = torch.randn(nq, 768, device=dev, generator=g)
for _ in range(20):
    ivf.search(q, k, nprobe)
torch.cuda.synchronize()
