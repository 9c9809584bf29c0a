"""Everything evaluate/run_repconc_eval.py:16-24 imports from the reference's models/repconc/evaluate_repconc.py, under
the same names: the argument dataclasses, `RepCONCEvaluater`, `encode_corpus` / `encode_query`, and the index helpers
`initialize_index / add_docs / from_pq_to_ivfpq / load_index_to_gpu / search / batch_search` on top of
repconc_amd.index.PQIndex.  No Faiss import anywhere.

`RepCONCEvaluater` keeps the reference's constructor and `.predict(dataset).predictions` contract
(evaluate_repconc.py:45-75,147-177) but is a plain batched loop, not an HF-Trainer subclass (the Trainer internals the
reference reaches into — `_prepare_inputs`, `autocast_smart_context_manager` — changed in transformers 5.x, SURVEY
Appendix C).  With `output_format="code"` the encoder output goes rotate -> `rc_pq_assign_nearest` -> uint8 codes on the
device (no int64 codes, no fp32 embeddings leaving the GPU); under torch.distributed every rank encodes a
contiguous share of the dataset and the shares are all-gathered once.
"""
from __future__ import annotations

import logging
import math
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Dict, Optional, Union

import numpy as np
import torch
import torch.distributed as dist
from transformers import TrainingArguments

from ... import ops
from ...index import METRIC_INNER_PRODUCT, PQIndex
from ...utils.eval_utils import TextDataset, get_collator_func

logger = logging.getLogger(__name__)


@dataclass
class ModelArguments:
    """evaluate_repconc.py:21-31."""
    model_name_or_path: str = field(default=None)
    doc_encoder_path: str = field(default=None)
    query_encoder_path: str = field(default=None)
    max_seq_length: int = field(default=None)

    def __post_init__(self):
        if self.model_name_or_path is not None:
            assert self.doc_encoder_path is None and self.query_encoder_path is None
            self.doc_encoder_path = self.query_encoder_path = self.model_name_or_path


@dataclass
class EvalArguments(TrainingArguments):
    """evaluate_repconc.py:33-43 (`threads` is accepted for command-line compatibility: there is no Faiss-OpenMP pool
    to size, the scan runs on the GPU; `cpu_search` is refused at search time — there is no CPU path)."""
    topk: int = field(default=1000)
    threads: int = field(default=1)
    search_batch: int = field(default=1200)
    cpu_search: bool = field(default=False)
    remove_unused_columns: Optional[bool] = field(default=False)


class RepCONCEvaluater:
    """`RepCONCEvaluater(output_format, model=..., args=..., data_collator=..., tokenizer=...).predict(dataset)`
    -> object with `.predictions` (numpy: uint8 codes [n, M] or fp32 continuous embeddings [n, D]) in dataset order."""

    def __init__(self, output_format: str, model=None, args=None, data_collator=None, tokenizer=None, **_ignored):
        assert output_format in ("code", "continuous_embedding")
        self.output_format, self.model, self.args = output_format, model, args
        self.data_collator, self.tokenizer = data_collator, tokenizer

    @torch.no_grad()
    def prediction_step(self, model, inputs, prediction_loss_only=False, ignore_keys=None):
        """One batch -> (None, codes uint8 | continuous embeddings, text_ids).  evaluate_repconc.py:51-75."""
        assert not prediction_loss_only and ignore_keys is None
        dev = model.centroids.device
        inputs = {k: v.to(dev, non_blocking=True) for k, v in inputs.items()}
        text_ids = inputs.pop("text_ids", None)
        amp = bool(getattr(self.args, "fp16", False) or getattr(self.args, "bf16", False))
        dtype = torch.bfloat16 if getattr(self.args, "bf16", False) else torch.float16
        with torch.autocast("cuda", dtype=dtype, enabled=amp):
            out = model(**inputs)                                   # continuous (rotated) embeddings only
        if self.output_format == "code":
            if model.use_constraint:                                # as the reference: whatever quantize() is set to
                logits = model.quantize(out.continuous_embeds).to(torch.uint8)
            else:                                                   # index build: nearest codes straight to uint8
                logits = ops.assign_nearest(out.continuous_embeds, model.centroids, torch.uint8)
        else:
            logits = out.continuous_embeds.detach().float()
        return None, logits, text_ids

    def predict(self, dataset):
        model = self.model
        model.eval()
        n = len(dataset)
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        bs = int(getattr(self.args, "per_device_eval_batch_size", 64) or 64)
        chunks = []
        for s in range(lo, hi, bs):
            batch = self.data_collator([dataset[i] for i in range(s, min(s + bs, hi))])
            chunks.append(self.prediction_step(model, batch)[1])
        dev = model.centroids.device
        width = model.config.MCQ_M if self.output_format == "code" else model.config.hidden_size
        dt = torch.uint8 if self.output_format == "code" else torch.float32
        local = torch.cat(chunks, 0) if chunks else torch.empty((0, width), dtype=dt, device=dev)
        if world > 1:                                               # one all-gather of the (padded) shares
            most = max((n * (r + 1)) // world - (n * r) // world for r in range(world))
            pad = torch.zeros((most, width), dtype=dt, device=dev)
            pad[: local.shape[0]] = local
            got = ops.all_gather(pad)                               # [world, most, width]: the handle's exchange layer
            ops.comm_check(dev)
            local = torch.cat([got[r][: (n * (r + 1)) // world - (n * r) // world] for r in range(world)], 0)
        return SimpleNamespace(predictions=local.cpu().numpy(), label_ids=None, metrics={})


def encode_corpus(corpus: Dict[Union[str, int], str], model, tokenizer, max_seq_length: int, eval_args):
    """Documents sorted longest first (padding efficiency), encoded to codes, appended to a fresh index.
    Returns (index, corpus_ids in index order).  evaluate_repconc.py:138-160."""
    corpus_ids = np.array(sorted(corpus, key=lambda k: len(corpus[k]), reverse=True))
    texts = [corpus[cid] for cid in corpus_ids]
    out = RepCONCEvaluater("code", model=model, args=eval_args, tokenizer=tokenizer,
                           data_collator=get_collator_func(tokenizer, max_seq_length, input_text_type="doc")
                           ).predict(TextDataset(texts, text_ids=list(range(len(texts)))))
    assert len(out.predictions) == len(corpus)
    index = initialize_index(model)
    add_docs(index, out.predictions)
    return index, corpus_ids


def encode_query(queries: Dict[int, str], model, tokenizer, max_seq_length: int, eval_args):
    """(query embeddings [nq, D] fp32 numpy, query ids sorted ascending).  evaluate_repconc.py:163-177."""
    query_ids = sorted(queries.keys())
    texts = [queries[q] for q in query_ids]
    out = RepCONCEvaluater("continuous_embedding", model=model, args=eval_args, tokenizer=tokenizer,
                           data_collator=get_collator_func(tokenizer, max_seq_length, input_text_type="query")
                           ).predict(TextDataset(texts, list(range(len(texts)))))
    assert len(out.predictions) == len(texts)
    return out.predictions, np.array(query_ids)


def initialize_index(model) -> PQIndex:
    """IndexPQ(D, M, 8, IP) holding the model's centroids.  evaluate_repconc.py:78-86."""
    D, M = model.config.hidden_size, model.config.MCQ_M
    assert model.config.MCQ_K == 256
    dev = model.centroids.device
    index = PQIndex(D, M, 8, METRIC_INNER_PRODUCT, device=dev if dev.type == "cuda" else None)
    index.set_centroids(model.centroids.data)
    return index


def add_docs(index: PQIndex, new_codes):
    """Append raw codes [n, M].  evaluate_repconc.py:89-98."""
    new_n = len(new_codes)
    assert tuple(new_codes.shape) == (new_n, index.pq.code_size)
    index.add_codes(new_codes)


def from_pq_to_ivfpq(indexpq: PQIndex) -> PQIndex:
    """The reference wraps the PQ index as an IVFPQ with ONE list and a zero coarse centroid
    (evaluate_repconc.py:101-118), which scores exactly like the flat index; nothing to convert."""
    return indexpq


def load_index_to_gpu(index: PQIndex, single_gpu_id=None, shard: bool = False, devices=None):
    """evaluate_repconc.py:121-135.  `single_gpu_id` given: the index on that device (moved if needed, :123-129).
    `single_gpu_id=None`: the reference's `index_cpu_to_all_gpus` with `co.shard = False` (:131-134) — a full copy on
    every visible GPU, each query batch split across the copies (`ReplicatedPQIndex`); `shard=True` row-shards the
    index instead (`ShardedPQIndex`).  With one visible GPU the index itself is returned.  `devices`
    overrides the device list (tests list one device twice)."""
    if single_gpu_id is None:
        from ...multi_index import ReplicatedPQIndex, ShardedPQIndex
        devs = list(devices) if devices is not None else list(range(torch.cuda.device_count()))
        if len(devs) <= 1:
            return index if not devs or index.device.index == devs[0] else load_index_to_gpu(index, devs[0])
        return (ShardedPQIndex if shard else ReplicatedPQIndex)(index, devs)
    if single_gpu_id is not None and index.device.index != single_gpu_id:
        dev = torch.device("cuda", single_gpu_id)
        moved = PQIndex(index.pq.d, index.pq.M, index.pq.nbits, index.metric_type, device=dev)
        moved.set_centroids(index.pq.centroids)
        moved.add_codes(index.codes)
        moved.id_offset = index.id_offset
        return moved
    return index


def search(query_ids: np.ndarray, query_embeds, corpus_ids: np.ndarray, index: PQIndex, topk: int):
    """evaluate_repconc.py:180-185."""
    topk_scores, topk_idx = index.search(query_embeds, topk)
    if isinstance(topk_idx, torch.Tensor):
        topk_idx, topk_scores = topk_idx.cpu().numpy(), topk_scores.cpu().numpy()
    topk_ids = np.asarray(corpus_ids)[topk_idx]
    assert len(query_ids) == len(topk_scores) == len(topk_ids)
    return topk_scores, topk_ids


def batch_search(query_ids: np.ndarray, query_embeds, corpus_ids: np.ndarray, index: PQIndex, topk: int,
                 batch_size: int):
    """evaluate_repconc.py:188-206 (np.array_split batching).  Every batch is enqueued before the first result is read
    (`search_async`), so the device runs the batches back to back; indexes without `search_async` (the multi-device
    wrappers, anything Faiss-shaped) take the reference's batch-by-batch loop."""
    if getattr(index, "whole_query_set", False):
        # a list-centric index (repconc_amd.ivf.IVFPQIndex) scans a probed cell once for every query that probes it: the more
        # queries in hand, the fuller its gather columns — all of them go in one call (the index chunks at 16 384 itself)
        return search(query_ids, query_embeds, corpus_ids, index, topk)
    iterations = max(1, math.ceil(len(query_ids) / batch_size))
    qid_parts, emb_parts = np.array_split(query_ids, iterations), np.array_split(query_embeds, iterations)
    if not hasattr(index, "search_async"):
        got = [search(qid_it, emb_it, corpus_ids, index, topk) for qid_it, emb_it in zip(qid_parts, emb_parts)]
        return np.concatenate([g[0] for g in got], axis=0), np.concatenate([g[1] for g in got], axis=0)
    pending = [index.search_async(emb_it, topk) for emb_it in emb_parts]
    all_scores, all_ids = [], []
    ids_table = np.asarray(corpus_ids)
    for qid_it, fin in zip(qid_parts, pending):
        topk_scores, topk_idx = fin()
        if isinstance(topk_idx, torch.Tensor):
            topk_idx, topk_scores = topk_idx.cpu().numpy(), topk_scores.cpu().numpy()
        assert len(qid_it) == len(topk_scores) == len(topk_idx)
        all_scores.append(topk_scores)
        all_ids.append(ids_table[topk_idx])
    return np.concatenate(all_scores, axis=0), np.concatenate(all_ids, axis=0)
