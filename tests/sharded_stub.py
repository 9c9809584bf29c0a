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

    @staticmethod
    def _ranksum(rows_prev):
        rp = rows_prev if rows_prev.dim() == 3 else rows_prev.unsqueeze(0)
        tot = np.zeros(tuple(rp.shape[1:]))
        for r in range(rp.shape[0]):                                            # rank order
            tot = tot + rp[r].numpy()
        return tot

    def sweep(self, eps, t, rows_prev):
        if self.L is None:
            self.L = -(self.d.numpy().astype(np.float64)) / eps
        if t == 0:
            return torch.from_numpy(np.exp(self.L).sum(axis=1))                # [M,K]
        self.f = (self.f if t > 1 else 0.0) - np.log(self._ranksum(rows_prev))
        if t > 1:
            self.g = self.g - np.log(self.colsum)
        w = np.exp(self.L + self.f[:, None, :] + self.g[:, :, None])
        self.colsum = w.sum(axis=2)
        return torch.from_numpy((w / self.colsum[:, :, None]).sum(axis=1))

    def argmax(self, eps, t, rows_prev, dtype=torch.int64):
        f = (self.f if t > 1 else 0.0) - np.log(self._ranksum(rows_prev))
        return torch.from_numpy(np.argmax(self.L + f[:, None, :], axis=-1).T.copy()).to(dtype)


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
