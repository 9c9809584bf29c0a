#!/usr/bin/env python3
"""Quick ADC timing (development tool, GPU): 1200-query batches over the 8.84 M-row index through PQIndex (image kept by
the index) for M in argv (default 48 96), k in {1000, 10}; RC_ADC_OLD_SCREEN=1 times the round-1 screen.  Prints the
scan-kernel time from the HIP-event hook as well."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import _lib, ops  # noqa: E402
from repconc_amd.index import PQIndex  # noqa: E402

dev = "cuda:0"
N, D, QB = 8841823, 768, 1200
Ms = [int(a) for a in sys.argv[1:]] or [48, 96]
g = torch.Generator(device=dev).manual_seed(20222)
q = torch.randn(3 * QB, D, device=dev, generator=g)
lib, h = _lib.load(), _lib.handle(0)
for M in Ms:
    C = torch.randn(M, 256, D // M, device=dev, generator=g)
    idx = PQIndex(D, M)
    idx.set_centroids(C)
    idx.add_codes(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g))
    for k in (1000, 10):
        idx.search(q[:QB], k)
        torch.cuda.synchronize()
        lib.rc_profile_enable(h, 1)
        t0 = time.perf_counter()
        for s in range(0, 3 * QB, QB):
            idx.search(q[s:s + QB], k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lib.rc_profile_enable(h, 0)
        cnt, ms = ctypes.c_int(0), ctypes.c_double(0)
        lib.rc_profile_collect(h, 1, ctypes.byref(cnt), ctypes.byref(ms))
        print(f"M={M} k={k:4d} old_screen={os.environ.get('RC_ADC_OLD_SCREEN', '0')}: {3*QB/dt/1e3:7.1f} k QPS, "
              f"{dt/3*1e3:6.2f} ms per batch, scan kernel {ms.value/max(cnt.value,1):6.2f} ms x{cnt.value}", flush=True)
    del idx
