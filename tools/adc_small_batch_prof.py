"""Small-batch flat search (JPQ training steps, finetune_jpq.py:176: a few hundred queries, k = 200): the command
tools/adc_small_batch_prof.sh profiles — 40 searches of argv[1] (default 128) queries, k = argv[2] (default 200), M = 48, 8.84 M rows."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd.index import PQIndex
dev = "cuda:0"
N, D, M = 8841823, 768, 48
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 128
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = torch.Generator(device=dev).manual_seed(20222)
idx = PQIndex(D, M)
idx.set_centroids(torch.randn(M, 256, D // M, device=dev, generator=g))
idx.add_codes(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g))
q = torch.randn(40 * nq, D, device=dev, generator=g)
idx.search(q[:nq], k); torch.cuda.synchronize()
t0 = time.perf_counter()
for b in range(40):
    idx.search(q[b * nq:(b + 1) * nq], k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 40
print(f"nq={nq} k={k}: {dt*1e3:.3f} ms per search = {nq/dt/1e3:.1f} k QPS", flush=True)
