#!/usr/bin/env python3
"""Index-build micro-benchmark (development tool, GPU): nearest-code assignment of N x 768 vectors.
    python tools/nearest_bench.py [N] [M]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import _lib, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
M = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = "cuda:0"
x = torch.randn(N, 768, device=dev)
C = x[torch.randperm(N, device=dev)[:256]].reshape(256, M, 768 // M).transpose(0, 1).contiguous()
lib, h = _lib.load(), _lib.handle(0)
flops = N * 768 * 256 * 3
ref = None
for method in ("exact", "mfma"):
    ops.assign_nearest(x[:4096], C, torch.uint8, method=method)
    torch.cuda.synchronize()
    lib.rc_profile_enable(h, 1)
    st = {}
    t0 = time.perf_counter()
    codes = ops.assign_nearest(x, C, torch.uint8, method=method, stats=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.rc_profile_enable(h, 0)
    cnt, ms = ctypes.c_int(0), ctypes.c_double(0)
    lib.rc_profile_collect(h, 2, ctypes.byref(cnt), ctypes.byref(ms))
    same = "" if ref is None else f", equal to exact: {bool(torch.equal(ref, codes))}"
    ref = codes if ref is None else ref
    print(f"N={N} M={M} {method}: kernels {ms.value:.2f} ms ({N/ms.value*1e3/1e6:.2f} M vectors/s, "
          f"{flops/ms.value/1e9:.1f} T fp32-op/s, {N*(768*4+M)/ms.value/1e6:.0f} GB/s), wall {dt*1e3:.2f} ms, "
          f"doubtful {st.get('doubtful')}{same}")
