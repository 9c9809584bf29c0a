"""numpy stand-in for the per-rank HIP stages, so the multi-rank choreography of
repconc_amd.sharded can run under gloo on CPU.  TEST-ONLY: it restates the stage contract of
include/repconc_hip.h (rc_pq_dist_table / rc_pq_centre / rc_sk_pass / rc_sk_update / rc_sk_argmax)
on top of the oracle; the product never imports it."""
import numpy as np
import torch

from oracle import pq_oracle


class _State:
    def __init__(self, d):
        self.d = d                                   # torch fp32 [M,B,K] (centred)
        self.L = None
        M, B, K = d.shape
        self.f = np.zeros((M, K))
        self.g = np.zeros((M, B))
        self.colsum = None
        self.flags = torch.zeros(1, dtype=torch.int32)

    def sweep(self, eps, first):
        if self.L is None:
            self.L = -(self.d.numpy().astype(np.float64)) / eps
        if first:
            rows = np.exp(self.L).sum(axis=1)                                   # [M,K]
        else:
            w = np.exp(self.L + self.f[:, None, :] + self.g[:, :, None])
            self.colsum = w.sum(axis=2)
            rows = (w / self.colsum[:, :, None]).sum(axis=1)
        return torch.from_numpy(rows)

    def update(self, rows_all, first):
        tot = np.zeros_like(self.f)
        for r in range(rows_all.shape[0]):                                      # rank order
            tot = tot + rows_all[r].numpy()
        if first:
            self.f = -np.log(tot)
        else:
            self.g = self.g - np.log(self.colsum)
            self.f = self.f - np.log(tot)

    def argmax(self, eps, dtype=torch.int64):
        return torch.from_numpy(np.argmax(self.L + self.f[:, None, :], axis=-1).T.copy()).to(dtype)


class NumpyStages:
    def dist_table(self, x, centroids):
        d = pq_oracle.dist_table(x.numpy(), centroids.numpy())
        mx, mn = pq_oracle.minmax_per_m(d)
        return torch.from_numpy(d), torch.from_numpy(np.concatenate([mx, mn]))

    def centre_(self, d, minmax):
        M = d.shape[0]
        mm = minmax.numpy()
        d.copy_(torch.from_numpy(pq_oracle.centre(d.numpy(), mm[:M], mm[M:])))
        return d

    def state(self, d):
        return _State(d)
