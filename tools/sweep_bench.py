#!/usr/bin/env python3
"""Micro-benchmark of the Sinkhorn sweep kernel alone (development tool, GPU only).
    RC_SK_VARIANT=0|1 python tools/sweep_bench.py [B] [M] [sweeps]
Prints avg sweep time (HIP events around each launch) and algorithmic GB/s."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
M = int(sys.argv[2]) if len(sys.argv) > 2 else 48
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = "cuda:0"
if 768 % M == 0 and (768 // M) in ops.SUPPORTED_DSUB and not os.environ.get("SYNTH_D"):
    x = torch.randn(B, 768, device=dev)
    C = x[torch.randperm(B, device=dev)[:256]].reshape(256, M, 768 // M).transpose(0, 1).contiguous()
    d, mm = ops.dist_table(x, C)
    ops.centre_(d, mm)
else:   # synthetic centred table, any M (cache-residency experiments)
    d = (torch.randn(M, B, 256, device=dev) * 0.22).clamp_(-1, 1)
st = ops.SinkhornState(d)
rows = st.sweep(0.003, 0, None)
t = 1
for _ in range(4):
    rows = st.sweep(0.003, t, rows); t += 1
torch.cuda.synchronize()
lib, h = _lib.load(), _lib.handle(0)
lib.rc_profile_enable(h, 1)
for _ in range(n):
    rows = st.sweep(0.003, t, rows); t += 1
torch.cuda.synchronize()
lib.rc_profile_enable(h, 0)
cnt, ms = ctypes.c_int(0), ctypes.c_double(0)
lib.rc_profile_collect(h, 0, ctypes.byref(cnt), ctypes.byref(ms))
avg = ms.value / cnt.value
print(f"variant={os.environ.get('RC_SK_VARIANT', 'default')} B={B} M={M}: {avg*1e3:.1f} us/sweep, "
      f"{B*M*256*4/avg/1e6:.0f} GB/s algorithmic, flags={int(st.flags.item())}")
