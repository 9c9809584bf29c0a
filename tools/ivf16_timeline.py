#!/usr/bin/env python3
"""Per-wave timeline of the 16-query IVF screen (development tool, GPU box).  Needs a library built with -DRC_IVF_TRACE:
    tools/mkvar.sh trace16 ivf_lists.hip -DRC_IVF_TRACE          (here)
    REPCONC_HIP_LIB=$PWD/build/var/trace16.so python tools/ivf16_timeline.py [nprobe=128] [nq=1200]     (GPU box)
Stamps per wave and stage (first 64 stages of every block): 0 = after the stage's barrier, 1 = gathers done (gathering waves) /
tables stored (loader waves), 2 = end of the stage's work (epilogue + code hand-over / next phase requested)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import _lib  # noqa: E402
from repconc_amd.ivf import IVFPQIndex  # noqa: E402

dev = "cuda:0"
N, M, nlist, k = 8841823, 96, 5000, 1000
nprobe = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
g = torch.Generator(device=dev).manual_seed(1)
ivf = IVFPQIndex(768, M, nlist, device=dev)
ivf.set_centroids(torch.randn(M, 256, 768 // M, device=dev, generator=g))
ivf.coarse = torch.randn(nlist, 768, device=dev, generator=g)
ivf.set_lists(torch.randint(0, 256, (N, M), dtype=torch.uint8, device=dev, generator=g), torch.randint(0, nlist, (N,), device=dev, generator=g))
q = torch.randn(nq, 768, device=dev, generator=g)
for _ in range(3):
    ivf.search(q, k, nprobe, method="lists16")
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(256 * 64 * 16 * 4, dtype=np.uint64)
lib.rc_debug_ivfs16_trace.argtypes = [ctypes.c_void_p]
lib.rc_debug_ivfs16_trace(buf.ctypes.data)
t = buf.reshape(256, 64, 16, 4).astype(np.int64)[:, 12:60] * 0.01           # stages 12 .. 59 of every block; us (ticks of 10 ns)
gw, lw = t[:, :, :12], t[:, :, 12:]
start = t[..., 0].max(axis=2)                                               # the barrier releases everybody at about the same time
period = np.diff(start, axis=1)
print(f"16-query IVF screen, M = {M}, nprobe = {nprobe}, {nq} queries: means over 256 blocks x 48 stages (12 gathering + 4 loader waves)")
print(f"  stage period {period.mean():.2f} us (every sixth stage carries the survivor pass)")
gat = gw[..., 1] - gw[..., 0]
rest = gw[..., 2] - gw[..., 1]
print(f"  gathering waves: gathers {gat.mean():.2f} us (fastest wave of a stage {gat.min(-1).mean():.2f}, slowest {gat.max(-1).mean():.2f}); "
      f"then epilogue / code hand-over {rest.mean():.2f} us (slowest {rest.max(-1).mean():.2f}); "
      f"slowest wave done {(gw[..., 2].max(-1) - start).mean():.2f} us after the barrier")
sto = lw[..., 1] - lw[..., 0]
req = lw[..., 2] - lw[..., 1]
print(f"  loader waves: wait for the requested phase + transposition + stores {sto.mean():.2f} us (slowest {sto.max(-1).mean():.2f}); "
      f"next task / request {req.mean():.2f} us (slowest {req.max(-1).mean():.2f}); "
      f"slowest loader done {(lw[..., 2].max(-1) - start).mean():.2f} us after the barrier")
done_all = t[..., 2].max(axis=2)
print(f"  barrier: last wave arrives {(done_all - start).mean():.2f} us after the previous release; release-to-release {period.mean():.2f} us "
      f"=> {period.mean() - (done_all[:, :-1] - start[:, :-1]).mean():.2f} us between the last arrival and the next stage's first stamp")
skew = t[..., 0].max(axis=2) - t[..., 0].min(axis=2)
print(f"  spread of the waves' first stamps after a barrier: {skew.mean():.2f} us")
