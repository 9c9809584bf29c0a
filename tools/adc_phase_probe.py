#!/usr/bin/env python3
"""Timing probe for the q16 screen's phase changes (development tool, GPU): an index whose three table phases are IDENTICAL
(centroids and query sub-vectors repeat with period 16 sub-quantisers), so A/B libraries that skip the refill or the barrier
(DBG_NOREFILL / DBG_NOBARRIER: wrong in general) still return the right answer here and only their time differs."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import _lib  # noqa: E402
from repconc_amd.index import PQIndex  # noqa: E402
dev = "cuda:0"
N, D, QB, M = 8841823, 768, 1200, int(sys.argv[1]) if len(sys.argv) > 1 else 48
g = torch.Generator(device=dev).manual_seed(20222)
ds = D // M
C16 = torch.randn(16, 256, ds, device=dev, generator=g)
C = C16.repeat(M // 16, 1, 1).contiguous()
q16 = torch.randn(QB, 16, ds, device=dev, generator=g)
q = q16.repeat(1, M // 16, 1).reshape(QB, D).contiguous()
lib, h = _lib.load(), _lib.handle(0)
idx = PQIndex(D, M)
idx.set_centroids(C)
idx.add_codes(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g))
ref = idx.search(q, 1000)
torch.cuda.synchronize()
lib.rc_profile_enable(h, 1)
t0 = time.perf_counter()
for _ in range(3):
    out = idx.search(q, 1000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.rc_profile_enable(h, 0)
cnt, ms = ctypes.c_int(0), ctypes.c_double(0)
lib.rc_profile_collect(h, 1, ctypes.byref(cnt), ctypes.byref(ms))
print(f"M={M}: {dt/3*1e3:6.2f} ms per batch, scan kernel {ms.value/max(cnt.value,1):6.2f} ms x{cnt.value}", flush=True)
