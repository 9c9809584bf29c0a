#!/usr/bin/env python3
"""Per-wave timeline of the pipelined IVF screen (development tool, GPU box): builds a variant of the library with
-DRC_IVF_TRACE (wall-clock stamps at the stage boundaries of the 8-QUERY screen, csrc/ivf_lists.hip; the 16-query
screen has tools/ivf16_timeline.py), runs the BASELINE configs[3] shape at
nprobe = argv[1] (default 128) and prints, per table phase, what the gathering and the loader waves spend where.
    python tools/ivf_timeline.py [nprobe]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, "repconc_amd", "lib")
VAR = os.path.join(LIBDIR, "librepconc_hip_trace.so")


def build_variant():
    from repconc_amd import build as b
    b.build(verbose=False)
    obj = os.path.join(LIBDIR, "ivf_lists_trace.o")
    subprocess.run([b._hipcc(), *b.FLAGS, "-DRC_IVF_TRACE", "-c", os.path.join(b.CSRC, "ivf_lists.hip"), "-o", obj], check=True)
    objs = [os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in b.SOURCES if s != "ivf_lists.hip"] + [obj]
    subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", VAR], check=True)


if os.environ.get("REPCONC_HIP_LIB") != VAR:
    build_variant()
    os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, REPCONC_HIP_LIB=VAR))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from repconc_amd import _lib  # noqa: E402
from repconc_amd.ivf import IVFPQIndex  # noqa: E402

dev = "cuda:0"
N, M, nlist, nq, k = 8841823, 96, 5000, 1200, 1000
nprobe = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator(device=dev).manual_seed(1)
ivf = IVFPQIndex(768, M, nlist, device=dev)
ivf.set_centroids(torch.randn(M, 256, 768 // M, device=dev, generator=g))
ivf.coarse = torch.randn(nlist, 768, device=dev, generator=g)
ivf.set_lists(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g), torch.randint(0, nlist, (N,), device=dev, generator=g))
q = torch.randn(nq, 768, device=dev, generator=g)
for _ in range(3):
    ivf.search(q, k, nprobe, method="lists8")
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(256 * 8 * 3 * 16 * 4, dtype=np.uint64)
lib.rc_debug_ivfs_trace.argtypes = [ctypes.c_void_p]
lib.rc_debug_ivfs_trace(buf.ctypes.data)
t = buf.reshape(256, 8, 3, 16, 4).astype(np.int64)[:, 2:7]          # tasks 2 .. 6 of every block; ticks of 10 ns
print(f"IVF screen timeline, M = {M}, nprobe = {nprobe}: means over 256 blocks x 5 tasks (12 gathering + 4 loader waves)")
for P in range(3):
    x = t[:, :, P]
    gw, lw = x[:, :, :12], x[:, :, 12:]
    gat = (gw[..., 1] - gw[..., 0]) * 0.01
    rest = (gw[..., 2] - gw[..., 1]) * 0.01
    fill = (lw[..., 2] - lw[..., 0]) * 0.01
    start = x[..., 0].min(axis=-1, keepdims=True)
    print(f"  phase {P}: gathers {gat.mean():.2f} us (fastest wave {gat.min(-1).mean():.2f}, slowest {gat.max(-1).mean():.2f}); "
          f"then code loads{' + survivor pass' if P == 2 else ''} {rest.mean():.2f} us (slowest {rest.max(-1).mean():.2f}); "
          f"slowest gathering wave done {((gw[..., 2] - start) * 0.01).max(-1).mean():.2f} us after the barrier; "
          f"loader waves {fill.mean():.2f} us (slowest {fill.max(-1).mean():.2f})")
print(f"  stage period {((t[:, :, 1, 0, 0] - t[:, :, 0, 0, 0]) * 0.01).mean():.2f} us, task period "
      f"{((t[:, 1:, 0, 0, 0] - t[:, :-1, 0, 0, 0]) * 0.01).mean():.2f} us")
