"""Index build / search glue with the reference's function names
(models/repconc/evaluate_repconc.py:78-135,180-206), on top of repconc_amd.index.PQIndex.

The HF-Trainer based `RepCONCEvaluater`, `encode_corpus` and `encode_query` of the reference are
harness code outside the hot path and are not re-implemented; their outputs
(uint8 codes [N,M] in corpus order, fp32 query embeddings) are exactly what these functions take.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ...index import METRIC_INNER_PRODUCT, PQIndex


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


def load_index_to_gpu(index: PQIndex, single_gpu_id=None) -> PQIndex:
    """evaluate_repconc.py:121-135.  The index is device resident from the start; with
    `single_gpu_id` it is moved to that device.  (Multi-GPU replication is per process: one
    process per GPU each holds — or row-shards — the index, see repconc_amd.sharded_search.)"""
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
    """evaluate_repconc.py:188-206 (np.array_split batching)."""
    iterations = max(1, math.ceil(len(query_ids) / batch_size))
    all_scores, all_ids = [], []
    for qid_it, emb_it in zip(np.array_split(query_ids, iterations), np.array_split(query_embeds, iterations)):
        s, i = search(qid_it, emb_it, corpus_ids, index, topk)
        all_scores.append(s)
        all_ids.append(i)
    return np.concatenate(all_scores, axis=0), np.concatenate(all_ids, axis=0)
