"""Stage-1 RepCONC training step without the HF-Trainer / GradCache machinery (first slice).

What the reference does per step (models/repconc/finetune_repconc.py:245-396, Appendix C of SURVEY.md):
  1. encode queries, positives and hard negatives WITHOUT grad, in chunks of `cache_chunk_size` texts (:312-314);
  2. constrained quantisation of cat(pos, neg) — the hot path (`model.quantize`, :317-318) — and decode (:321-322);
  3. contrastive loss on (query, QUANTISED docs) with false-negative / duplicate masks and optional dynamic hard
     negatives (:398-451); backward only to the representations -> cached gradients g (:340-341);
  4. second forward WITH grad, chunk by chunk, same dropout masks; the cached g is applied as a surrogate
     <g, x> to the continuous embedding (straight-through) and, for documents, <g, decode(codes)> (-> centroids) plus
     `mse_loss_weight * mean ||decode(codes) - x||^2` (:346-396).
This module restates that recipe, including the cross-rank gather of representations (:296-303,:331-335) when
torch.distributed is initialised, so that the kernels behind `quantize` / `decode` can be exercised inside a real optimisation step.  It is harness
code: all arithmetic on the hot path is still `repconc_amd.ops`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F


@dataclass
class Stage1Config:
    cache_chunk_size: int = 64                      # 7_run_conc_train.sh: --cache_chunk_size 64
    mse_loss_weight: float = 1e-4
    temperature: float = 1.0
    dynamic_topk_hard_negative: Optional[int] = None


def _chunks(td: Dict[str, torch.Tensor], n: int) -> List[Dict[str, torch.Tensor]]:
    keys = list(td)
    parts = [td[k].split(n) for k in keys]
    return [dict(zip(keys, p)) for p in zip(*parts)]


class _RngReplay:
    """Remember the RNG state in front of a chunk's forward so the second pass sees the same dropout masks."""

    def __init__(self, device):
        self.device = device
        self.cpu = torch.get_rng_state()
        self.cuda = torch.cuda.get_rng_state(device) if device.type == "cuda" else None

    def __enter__(self):
        self._keep_cpu = torch.get_rng_state()
        self._keep_cuda = torch.cuda.get_rng_state(self.device) if self.cuda is not None else None
        torch.set_rng_state(self.cpu)
        if self.cuda is not None:
            torch.cuda.set_rng_state(self.cuda, self.device)

    def __exit__(self, *exc):
        torch.set_rng_state(self._keep_cpu)
        if self._keep_cuda is not None:
            torch.cuda.set_rng_state(self._keep_cuda, self.device)


@torch.no_grad()
def _encode_no_grad(model, chunks):
    reps, states = [], []
    dev = next(model.parameters()).device
    for ch in chunks:
        states.append(_RngReplay(dev))
        reps.append(model(**ch).continuous_embeds)
    return torch.cat(reps, 0), states


def contrastive_loss(query_embeds, doc_embeds, qids, docids, qrels, cfg: Stage1Config, metric: str, M: int):
    """In-batch softmax over all documents; labels on the diagonal (finetune_repconc.py:398-431)."""
    nq = query_embeds.shape[0]
    labels = torch.arange(nq, device=query_embeds.device)
    # other relevant documents of a query are not negatives (:433-440)
    false_neg = torch.zeros((nq, docids.shape[0]), dtype=torch.bool, device=docids.device)
    for i, qid in enumerate(qids.tolist()):
        for d in qrels.get(int(qid), ()):
            false_neg[i] |= docids == d
    false_neg.fill_diagonal_(False)
    # a document that already occurred earlier in the batch is dropped as a negative (:442-451)
    dup = torch.triu(docids[:, None] == docids[None, :], diagonal=1).any(dim=0, keepdim=True).repeat(nq, 1)
    dup.fill_diagonal_(False)
    sim = query_embeds @ doc_embeds.T
    if metric == "METRIC_CENTROID_COS":
        sim = sim / M
    if cfg.temperature != 1:
        sim = sim / cfg.temperature
    sim = sim - 10000.0 * (false_neg | dup).float()
    if cfg.dynamic_topk_hard_negative:
        keep_out = torch.ones_like(sim)
        neg_sim = sim.detach().clone()
        neg_sim.scatter_(1, labels[:, None], -10000.0)
        keep_out.scatter_(1, torch.topk(neg_sim, cfg.dynamic_topk_hard_negative).indices, 0)
        keep_out.scatter_(1, labels[:, None], 0)
        sim = sim - 10000.0 * keep_out
    return F.cross_entropy(sim, labels)


def _all_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate every rank's rows in rank order (the reference's `gather_tensors`, finetune_repconc.py:297-300)."""
    world = dist.get_world_size(group)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.cat(parts, 0)


def global_loss_and_local_grads(q_rep, pos_quant, neg_quant, qids, pos_docids, neg_docids, qrels, cfg: "Stage1Config",
                                metric, M: int, group=None):
    """Loss of the GLOBAL batch and its gradient w.r.t. this rank's rows (finetune_repconc.py:296-303,325-341).

    With torch.distributed initialised every rank contributes its query / quantised-document representations (and ids);
    the contrastive loss runs over all queries x all documents (gathered positives first, then gathered negatives, as
    the reference stacks them), and the slices of the gradient cache that belong to the local rows are returned:
    (loss, g_query, g_pos, g_neg|None).  Equal row counts per rank, as the reference's DistributedSampler gives."""
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if multi else 0
    g = (lambda t: _all_rows(t, group)) if multi else (lambda t: t)
    nq, np_ = q_rep.shape[0], pos_quant.shape[0]
    nn_ = 0 if neg_quant is None else neg_quant.shape[0]
    q_leaf = g(q_rep.detach()).requires_grad_(True)
    docs = g(pos_quant.detach())
    docids = g(pos_docids)
    if neg_quant is not None:
        docs = torch.cat([docs, g(neg_quant.detach())], 0)
        docids = torch.cat([docids, g(neg_docids)])
    d_leaf = docs.requires_grad_(True)
    loss = contrastive_loss(q_leaf, d_leaf, g(qids), docids, qrels, cfg, metric, M)
    loss.backward()
    world = dist.get_world_size(group) if multi else 1
    g_q = q_leaf.grad[rank * nq:(rank + 1) * nq]
    g_p = d_leaf.grad[rank * np_:(rank + 1) * np_]
    g_n = d_leaf.grad[world * np_ + rank * nn_: world * np_ + (rank + 1) * nn_] if neg_quant is not None else None
    return loss.detach(), g_q, g_p, g_n


def stage1_training_step(model, query_input, pos_doc_input, qids, pos_docids, qrels, cfg: Stage1Config,
                         neg_doc_input=None, neg_docids=None, group=None) -> float:
    """One forward/backward of stage-1 training; gradients are left in `.grad` (call optimizer.step() after).
    Under torch.distributed the constrained quantisation already spans the global batch (`model.quantize`) and the loss
    is taken over the gathered representations (`global_loss_and_local_grads`)."""
    model.train()
    q_chunks = _chunks(query_input, cfg.cache_chunk_size)
    p_chunks = _chunks(pos_doc_input, cfg.cache_chunk_size)
    n_chunks = _chunks(neg_doc_input, cfg.cache_chunk_size) if neg_doc_input is not None else []
    q_rep, q_rng = _encode_no_grad(model, q_chunks)
    p_rep, p_rng = _encode_no_grad(model, p_chunks)
    if n_chunks:
        n_rep, n_rng = _encode_no_grad(model, n_chunks)
        docs = torch.cat([p_rep, n_rep], 0)
        docids = torch.cat([pos_docids, neg_docids])
    else:
        docs, docids = p_rep, pos_docids
    with torch.no_grad():
        codes = model.quantize(docs)                                     # the hot path (constrained if enabled)
        quantized = model.decode(codes)
    # loss on (query, quantised docs) of the global batch; gradients w.r.t. the local representations only
    np_ = p_rep.shape[0]
    loss, g_q, g_p, g_n = global_loss_and_local_grads(
        q_rep, quantized[:np_], quantized[np_:] if n_chunks else None, qids, pos_docids, neg_docids if n_chunks else None,
        qrels, cfg, getattr(model.config, "similarity_metric", None), model.config.MCQ_M, group)
    plan = [(q_chunks, g_q, q_rng, None), (p_chunks, g_p, p_rng, codes[:np_])]
    if n_chunks:
        plan.append((n_chunks, g_n, n_rng, codes[np_:]))
    for chunks, grads, rngs, doc_codes in plan:
        off = 0
        for ch, rng in zip(chunks, rngs):
            n = next(iter(ch.values())).shape[0]
            with rng:
                if doc_codes is None:
                    out = model(**ch)
                    obj = torch.dot(grads[off:off + n].flatten(), out.continuous_embeds.flatten())
                else:
                    out = model(discrete_codes=doc_codes[off:off + n], return_quantized_embedding=True, **ch)
                    g = grads[off:off + n].flatten()
                    obj = torch.dot(g, out.continuous_embeds.flatten()) + torch.dot(g, out.quantized_embeds.flatten())
                    obj = obj + ((out.quantized_embeds - out.continuous_embeds) ** 2).sum(-1).mean() * cfg.mse_loss_weight
            obj.backward()
            off += n
    return float(loss.item())


def allreduce_gradients_(model, group=None):
    """Average the parameter gradients over the ranks (what DistributedDataParallel does for the reference's trainer).
    `stage1_training_step` leaves LOCAL gradients of the global-batch loss in `.grad`: call this before
    `optimizer.step()` when training on more than one rank without a DDP wrapper."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    for p in model.parameters():
        if p.grad is not None:
            dist.all_reduce(p.grad, group=group)
            p.grad.div_(world)


def make_optimizer(model, lr: float = 2e-5, centroid_lr: float = 5e-4, weight_decay: float = 0.0):
    """AdamW with the reference's three groups: decay / no-decay encoder parameters and the centroids at their own
    learning rate, weight decay 0 (finetune_repconc.py:488-502)."""
    no_decay = ("bias", "LayerNorm.weight")
    enc = [(n, p) for n, p in model.named_parameters() if n != "centroids" and p.requires_grad]
    groups = [
        {"params": [p for n, p in enc if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
        {"params": [p for n, p in enc if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
        {"params": [model.centroids], "weight_decay": 0.0, "lr": centroid_lr},
    ]
    return torch.optim.AdamW(groups, lr=lr)
