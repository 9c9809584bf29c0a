#!/usr/bin/env python3
"""One-off full-size parity run (development tool, GPU box; ~1-2 min of host CPU): the constrained assignment of a whole
49 152 x 768 training batch (M = 48, eps 0.003, T = 100) against the C restatement of the reference, code by code, on
(a) clustered embeddings with sampled centroids and (b) the same embeddings with Lloyd-refined centroids.
Not part of the test suite (host time); the result is quoted in DESIGN.md §2."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle, pq_oracle, synth  # noqa: E402
from repconc_amd import ops  # noqa: E402

B, M = 49152, 48
x = synth.clustered_embeddings(777, B, n_clusters=512)
C0 = synth.sample_centroids(778, x, M)
dev = "cuda:0"
xt = torch.from_numpy(x).to(dev)
Ct = torch.from_numpy(C0).to(dev)
for _ in range(5):                                      # Lloyd refinement on the GPU kernels
    codes = ops.assign_nearest(xt, Ct, torch.uint8)
    sums, counts = ops.kmeans_stats(xt, codes)
    ops.kmeans_update_(sums, counts, Ct)
C1 = Ct.cpu().numpy()
for name, C in (("sampled centroids", C0), ("Lloyd-refined centroids", C1)):
    t0 = time.perf_counter()
    want, _ = c_oracle.quantize(x, C, True, 0.003, 100)
    tc = time.perf_counter() - t0
    got, flags = ops.assign_sinkhorn(xt, torch.from_numpy(C).to(dev), 0.003, 100, torch.uint8)
    got = got.cpu().numpy()
    mism = int((got != want).sum())
    hist = np.stack([np.bincount(got[:, m], minlength=256) for m in range(M)])
    print(f"{name}: {B} x {M} codes, mismatches vs C oracle: {mism}, flags {int(flags.item())}, "
          f"per-centroid counts {hist.min()}..{hist.max()} (ideal {B // 256}), oracle {tc:.1f} s on {c_oracle.num_threads()} threads",
          flush=True)
    near = ops.assign_nearest(xt, torch.from_numpy(C).to(dev), torch.uint8).cpu().numpy()
    print(f"   nearest codes: mismatches vs C oracle {int((near != c_oracle.quantize(x, C, False)[0]).sum())}", flush=True)
