"""Stage-2 (JPQ) model: query encoder + PQ centroids trained against a FIXED code index.

Mirror of the `JPQ` module of models/jpq/finetune_jpq.py:139-246 (constructor arguments, `forward(query_input_ids,
query_attention_mask, qids) -> {"loss"}`, `synchronize_model_index`, `normalize_centrodis`, `state_dict` /
`load_state_dict` delegating to the wrapped RepCONC) with the index side replaced:

  * the reference re-copies the centroid table into two Faiss indexes and re-clones the whole IVFPQ index (424 MB at
    M = 48) to the GPU after EVERY optimiser step (:209-214, callback :249-255).  Here `pq_index` is a
    `repconc_amd.index.PQIndex` whose codes never move; `synchronize_model_index` is `pq_index.set_centroids`
    (786 KB device copy), and the hard-negative search (:176) is `rc_adc_search` on CUDA tensors;
  * the int64 copy of all codes the reference registers as a buffer (:161-163, 3.4 GB at M = 48) is not made: rows of the
    resident uint8 codes are gathered and decoded directly (`rc_pq_decode` takes uint8), with the scatter-add backward
    into the centroids (`rc_pq_decode_bwd`).

The HF-Trainer subclass (`JPQFinetuner`), its callbacks and the dataset/collator are harness and stay out of scope;
`jpq_step_end` is what those callbacks do after each optimiser step.
"""
from __future__ import annotations

import random
from typing import Dict, List

import torch
from torch import nn

from ... import ops
from ...index import PQIndex
from ..repconc.modeling_repconc import RepCONC


class JPQ(nn.Module):
    def __init__(self, repconc: RepCONC, pq_index: PQIndex, qrels: Dict[int, List[int]], neg_top_k: int,
                 temperature: float, gpu_id=None):
        super().__init__()
        self.repconc = repconc
        self.qrels = qrels
        self.neg_top_k = neg_top_k
        self.temperature = temperature
        self.pq_index = pq_index
        self.gpu_id = gpu_id            # kept for signature compatibility: the index already lives on its device
        self.synchronize_model_index()

    @property
    def codes(self) -> torch.Tensor:
        """uint8 [N, M], the index's own resident codes (the reference keeps a second, int64 copy)."""
        return self.pq_index.codes

    def _decode_rows(self, pids: torch.Tensor) -> torch.Tensor:
        rows = self.pq_index.codes.index_select(0, pids.reshape(-1))
        return ops.decode(rows, self.repconc.centroids)            # differentiable w.r.t. the centroids

    def forward(self, query_input_ids: torch.Tensor, query_attention_mask: torch.Tensor, qids: torch.Tensor):
        query_embeds = self.repconc(query_input_ids, query_attention_mask, return_code=False,
                                    return_quantized_embedding=False).continuous_embeds          # [nq, D]
        with torch.no_grad():                                                                     # :176
            neg_pids = self.pq_index.search(query_embeds.detach().float().contiguous(), self.neg_top_k)[1]
        nq, k = neg_pids.shape
        # an index with fewer than neg_top_k rows pads the result with id -1: those slots decode row 0 and are then
        # pushed out of the softmax (the reference would index with -1, i.e. silently use the LAST row)
        empty = neg_pids < 0
        neg_doc_embeds = self._decode_rows(neg_pids.clamp_min(0)).reshape(nq, k, -1)
        neg_masks = self._compute_negative_mask(qids, neg_pids)
        query_negdoc_scores = (query_embeds.unsqueeze(1) * neg_doc_embeds).sum(-1) / self.temperature
        query_negdoc_scores = query_negdoc_scores.masked_fill(empty, -10000.0)
        pos_pids = torch.tensor([random.choice(self.qrels[int(q)]) for q in qids.tolist()], dtype=torch.int64,
                                device=neg_pids.device)
        rel_doc_embeds = self._decode_rows(pos_pids)
        query_reldoc_scores = (query_embeds * rel_doc_embeds).sum(-1, keepdim=True) / self.temperature
        loss = self.compute_loss(query_reldoc_scores, query_negdoc_scores, neg_masks)
        return {"loss": loss}

    @torch.no_grad()
    def _compute_negative_mask(self, qids: torch.Tensor, docids: torch.Tensor) -> torch.Tensor:
        """1.0 where a retrieved document is a labelled positive of its query (:196-207), [nq, k] fp32."""
        mask = torch.zeros(docids.shape, dtype=torch.bool, device=docids.device)
        for i, q in enumerate(qids.tolist()):
            rel = torch.tensor(self.qrels[int(q)], dtype=docids.dtype, device=docids.device)
            mask[i] = torch.isin(docids[i], rel)
        return mask.float()

    @torch.no_grad()
    def synchronize_model_index(self):
        self.pq_index.set_centroids(self.repconc.centroids.data)

    @torch.no_grad()
    def normalize_centrodis(self):
        self.repconc.normalize_centrodis()

    def state_dict(self, *args, **kwargs):
        return self.repconc.state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True):
        missing = self.repconc.load_state_dict(state_dict, strict)
        self.synchronize_model_index()
        return missing

    def compute_loss(self, query_reldoc_scores, query_negdoc_scores, neg_masks):
        """Cross-entropy of the positive against the retrieved documents (:232-246).  As in the reference the mask of
        false negatives is computed but does not enter the loss."""
        scores = torch.hstack((query_reldoc_scores, query_negdoc_scores))
        labels = torch.zeros(scores.size(0), dtype=torch.long, device=scores.device)
        return nn.functional.cross_entropy(scores, labels)


@torch.no_grad()
def jpq_step_end(model: JPQ):
    """What RepCONC_Norm_Centroid_Callback + JPQ_SyncIndex_Callback do after every optimiser step (:249-266)."""
    if getattr(model.repconc.config, "similarity_metric", None) == "METRIC_CENTROID_COS":
        model.normalize_centrodis()
    model.synchronize_model_index()
