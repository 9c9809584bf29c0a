"""Quantisation diagnostics logged during stage-1 training — models/repconc/finetune_repconc.py:580-613.

`eval_balance` needs the histogram of ONE sub-quantiser; the reference makes 256 `.sum().item()` round trips for
it (:590-592).  Here one `rc_code_hist` launch counts every sub-quantiser and a single 1 KB copy comes back."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import ops


def eval_balance(codes: torch.Tensor, distributed: bool, block_id: int):
    """finetune_repconc.py:580-597: |1 - n_k/(B/256)| mean and max over the 256 centroids of `block_id`,
    over the codes of every rank."""
    hist = ops.code_hist(codes)[block_id].to(torch.float64)
    n = codes.shape[0]
    if distributed:
        dist.all_reduce(hist)                               # == histogram of the all-gathered codes (:583-585)
        n *= dist.get_world_size()
    bal = (1.0 - hist / (n / 256)).abs().cpu().numpy()
    return {"avg_imbalance": round(float(np.mean(bal)), 3), "max_imbalance": round(float(np.max(bal)), 3)}


@torch.no_grad()
def test_quantize(continuous_embeddings: torch.Tensor, lm, local_rank: int, block_id: int = 0):
    """finetune_repconc.py:600-613: MSE (mean L2 norm of the residual, sic) and balance with and without the
    constraint; restores `lm.use_constraint`."""
    states = {}
    keep = lm.use_constraint
    for prefix, use_constraint in (("w/o_conc", False), ("w/_conc", True)):
        lm.use_constraint = use_constraint
        codes = lm.quantize(continuous_embeddings)
        quantized = lm.decode(codes)
        mse = ((quantized - continuous_embeddings) ** 2).sum(-1).sqrt().mean()
        states[f"{prefix}_mse"] = round(mse.item(), 3)
        states.update({f"{prefix}_{k}": v for k, v in eval_balance(codes, local_rank > -1, block_id).items()})
    lm.use_constraint = keep
    return states
