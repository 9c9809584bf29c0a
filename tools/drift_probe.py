#!/usr/bin/env python3
"""How much do the Sinkhorn row potentials still move after iteration t0, and how many matrix entries stay relevant?
(development probe for an active-set sweep).  For the bench batch: spread_t = max_k d_k - min_k d_k with
d_k = f^t_k - f^t0_k per sub-quantiser; active fraction = share of entries with L+f^t0 within Delta of their column max."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import ops  # noqa: E402

dev = "cuda:0"
B, M, eps, T = int(sys.argv[1]) if len(sys.argv) > 1 else 49152, 48, 0.003, 100
kind = sys.argv[2] if len(sys.argv) > 2 else "gauss"
if kind == "gauss":
    rng = np.random.default_rng(20220)
    xb = rng.standard_normal((B, 768), dtype=np.float32)
    cent = np.ascontiguousarray(xb[np.random.default_rng(20221).permutation(B)[:256]].reshape(256, M, 16).transpose(1, 0, 2))
else:
    from oracle import synth
    xb = synth.clustered_embeddings(7, B)
    cent = synth.sample_centroids(8, xb, M)
x, C = torch.from_numpy(xb).to(dev), torch.from_numpy(cent).to(dev)
d, mm = ops.dist_table(x, C)
ops.centre_(d, mm)
st = ops.SinkhornState(d)
rows = st.sweep(eps, 0, None)
fs = []
for t in range(1, T):
    rows = st.sweep(eps, t, rows)
    fs.append(st.f2[t & 1].clone())          # f_t
fs.append(st.potentials(T, rows))
fs = torch.stack(fs)                          # [T, M, K]: f_1 .. f_T
print(f"B={B} kind={kind}: f range per m (final): {float((fs[-1].max(1).values - fs[-1].min(1).values).mean()):.1f} nats")
for t0 in (3, 5, 10, 20, 30, 50):
    drift = fs[t0:] - fs[t0 - 1]              # f_t - f_t0 for t > t0
    spread = (drift.max(2).values - drift.min(2).values)          # [T-t0, M]
    worst = spread.max().item()
    print(f"  t0={t0:3d}: max over later t and m of spread = {worst:8.3f} nats; at T: mean {spread[-1].mean().item():.3f}")
# active fractions at t0 = 10 and 20
L = -(d.double()) / eps                        # [M,B,K] fp64: 4.8 GB at the full batch -> do a slice of m
for t0 in (10, 20):
    f0 = fs[t0 - 1]
    for Delta in (90, 100, 120, 150, 200):
        act = []
        for m in range(0, M, 12):
            s = L[m] + f0[m][None, :]
            gap = s.max(1, keepdim=True).values - s
            act.append((gap <= Delta).double().mean().item())
        print(f"  t0={t0}: Delta={Delta:4d} nats -> active fraction {np.mean(act):.4f} (max entries/column ~{256*np.mean(act):.1f})")
