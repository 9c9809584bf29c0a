"""torch-CPU restatement of the reference's constrained / nearest code assignment (TEST INFRASTRUCTURE, see
oracle/__init__.py; only tests/ and bench.py's cpu_baseline leg import it).

BASELINE.md 4.1 / SURVEY.md 8d plan the CPU baseline as "the build's own restatement with the reference's algorithmic shape:
torch-CPU, materialise the fp32 distance table, Sinkhorn in fp64 on the whole [M, K, B] matrix, 4096-row batches" — what
RepCONC.quantize (models/repconc/modeling_repconc.py:47-67) does on a CPU box, with torch's intra-op thread pool doing the
parallel work.  Written against the same tensor operations in the same order, so on the golden fixtures it returns the
reference's codes (tests/test_oracle_golden.py::test_torch_port_matches_reference).
"""
import torch


@torch.no_grad()
def quantize(x: torch.Tensor, centroids: torch.Tensor, use_constraint: bool, eps: float = 0.003, iters: int = 100) -> torch.Tensor:
    """x [B, D] fp32, centroids [M, K, dsub] fp32 (CPU) -> codes int64 [B, M]."""
    B = x.shape[0]
    M, K, dsub = centroids.shape
    # squared distances d[m, b, k], the [M, B, K, dsub] difference materialised like the reference (:50)
    diff = x.reshape(B, M, 1, dsub).transpose(0, 1) - centroids.unsqueeze(1)
    d = (diff ** 2).sum(-1)
    del diff
    if not use_constraint:
        return torch.argmin(d, dim=-1).t()                                   # :52
    # centring on the per-sub-quantiser range (:73-85)
    mx = d.max(-1).values.max(-1).values
    mn = d.min(-1).values.min(-1).values
    mid = (mx + mn) / 2
    amp = mx - mid + 1e-5
    d = (d - mid[:, None, None]) / amp[:, None, None]
    # Sinkhorn-Knopp on Q[m, k, b] = exp(-d / eps) in fp64, in place (:137-165)
    Q = torch.exp(-d.double().transpose(1, 2) / eps)
    del d
    Q /= Q.sum(-1, keepdim=True).sum(-2, keepdim=True)
    for _ in range(iters):
        Q /= torch.sum(Q, dim=2, keepdim=True)                               # rows: every centroid gets 1 / K
        Q /= K
        Q /= torch.sum(Q, dim=1, keepdim=True)                               # columns: every document gets 1 / B
        Q /= B
    Q *= B
    return torch.argmax(Q.transpose(1, 2), dim=-1).t()                       # :63, :66
