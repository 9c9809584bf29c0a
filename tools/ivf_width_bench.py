#!/usr/bin/env python3
"""8- vs 16-query list-centric IVF screen (development tool, GPU; round 6): M = 96, nlist = 5000, 8.84 M rows in uniformly
filled cells, 6 980- and 1 200-query calls, nprobe 8 / 32 / 128; whole search ms and the screen kernel's own time
(HIP-event hook), both widths, results compared.

    python tools/ivf_width_bench.py [M]
"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import _lib  # noqa: E402
from repconc_amd.ivf import IVFPQIndex  # noqa: E402

dev = "cuda:0"
N, M, nlist, k = 8841823, int(sys.argv[1]) if len(sys.argv) > 1 else 96, 5000, 1000
g = torch.Generator(device=dev).manual_seed(1)
codes = torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g)
cells = torch.randint(0, nlist, (N,), device=dev, generator=g)
ivf = IVFPQIndex(768, M, nlist, device=dev)
ivf.set_centroids(torch.randn(M, 256, 768 // M, device=dev, generator=g))
ivf.coarse = torch.randn(nlist, 768, device=dev, generator=g)
ivf.set_lists(codes, cells)
del codes, cells
lib, h = _lib.load(), _lib.handle(0)
for nq in (6980, 1200):
    q = torch.randn(nq, 768, device=dev, generator=g)
    for nprobe in (8, 32, 128):
        res = {}
        for method in ("lists8", "lists16"):
            for _ in range(2):
                s, i = ivf.search(q, k, nprobe, method=method)
            torch.cuda.synchronize()
            lib.rc_profile_enable(h, 1)
            t0 = time.perf_counter()
            for _ in range(3):
                s, i = ivf.search(q, k, nprobe, method=method)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            lib.rc_profile_enable(h, 0)
            cnt, ms = ctypes.c_int(0), ctypes.c_double(0)
            lib.rc_profile_collect(h, 1, ctypes.byref(cnt), ctypes.byref(ms))
            res[method] = (dt, ms.value / max(cnt.value, 1), s, i)
        same = torch.equal(res["lists8"][3], res["lists16"][3]) and torch.equal(res["lists8"][2], res["lists16"][2])
        print(f"M={M} nq={nq} nprobe={nprobe:3d} (queries per probed cell {nq*nprobe/nlist:5.1f}): "
              f"8-query {res['lists8'][0]*1e3:7.2f} ms (screen {res['lists8'][1]:6.3f}), "
              f"16-query {res['lists16'][0]*1e3:7.2f} ms (screen {res['lists16'][1]:6.3f}); "
              f"{nq/res['lists8'][0]/1e3:7.1f} -> {nq/res['lists16'][0]/1e3:7.1f} k QPS; identical results: {same}", flush=True)
