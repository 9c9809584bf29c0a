"""Corpus -> codes -> resident index without the fp32 round trip.

The reference runs `Trainer.predict`, gathers every rank's uint8 codes on EVERY rank, and each rank then builds the
whole index redundantly (models/repconc/evaluate_repconc.py:51-75,147-158).  Here a rank encodes its own share of the
corpus, the rotated embeddings go straight from the encoder's output tensor into `rc_pq_assign_nearest`, and the codes
are appended to that rank's `PQIndex` shard (`id_offset` = global position of its first row).  The shards are searched
with `repconc_amd.sharded_search.sharded_search` (or all-gathered once into a replica).
"""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

import torch

from . import ops
from .index import PQIndex


@torch.no_grad()
def encode_corpus_to_index(model, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], id_offset: int = 0,
                           index: Optional[PQIndex] = None) -> PQIndex:
    """`batches` yields (input_ids, attention_mask) on the model's device, in corpus order for this rank.
    Equivalent to `model(..., return_code=True)` + `add_docs` per batch (modeling_repconc.py:87-110 with
    use_constraint=False; evaluate_repconc.py:69,89-98), minus the int64 -> uint8 -> numpy -> Faiss hops."""
    dev = model.centroids.device
    if index is None:
        index = PQIndex(model.config.hidden_size, model.config.MCQ_M, 8, device=dev)
        index.set_centroids(model.centroids.data)
        index.id_offset = id_offset
    for input_ids, attention_mask in batches:
        out = model(input_ids=input_ids, attention_mask=attention_mask)          # continuous (rotated) embeddings only
        index.add_codes(ops.assign_nearest(out.continuous_embeds, model.centroids, torch.uint8))
    return index
