"""Data plumbing of the evaluation scripts, with the reference's names (utils/eval_utils.py): TSV loaders, the text
dataset / collator pair the encoders are fed from, and rank metrics.  Harness code around the hot path — kept small,
no third-party metric package (the reference scores with pytrec_eval, which is not installable offline; `mrr_at_k` /
`recall_at_k` restate the two numbers the recipes report: MRR@k = mean over queries of 1 / rank of the first relevant
hit within the top k, relevance >= 1, eval_utils.py:136-190)."""
from __future__ import annotations

import inspect
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset


def load_corpus(corpus_path: str, sep_token: str, verbose: bool = True) -> Dict[str, str]:
    """id \\t field [\\t field ...] per line; fields joined with the tokenizer's separator, 10 000 characters kept."""
    corpus = {}
    with open(corpus_path) as f:
        for line in f:
            parts = line.strip().split("\t")
            corpus[parts[0]] = sep_token.join(p.strip() for p in parts[1:])[:10000]
    return corpus


def load_queries(query_path: str) -> Dict[str, str]:
    queries = {}
    with open(query_path) as f:
        for line in f:
            qid, text = line.split("\t")
            queries[qid] = text
    return queries


class TextDataset(Dataset):
    """A list of texts, optionally paired with integer ids that travel through the collator as `text_ids`."""

    def __init__(self, text_lst: List[str], text_ids: Optional[List[int]] = None):
        assert text_ids is None or len(text_ids) == len(text_lst)
        self.text_lst, self.text_ids = text_lst, text_ids

    def __len__(self):
        return len(self.text_lst)

    def __getitem__(self, i):
        return self.text_lst[i] if self.text_ids is None else (self.text_ids[i], self.text_lst[i])


def get_collator_func(tokenizer, max_length: int, input_text_type: str):
    """Batch of texts (or (id, text) pairs) -> input_ids / attention_mask (/ text_ids).  Tokenizers whose __call__ takes
    `input_text_type` (the TCT-ColBERT recipe's) are told whether they see queries or documents."""
    try:
        extra = {"input_text_type": input_text_type} if "input_text_type" in inspect.signature(tokenizer.__call__).parameters else {}
    except (TypeError, ValueError):
        extra = {}

    def collate(batch):
        paired = isinstance(batch[0], tuple)
        texts = [b[1] for b in batch] if paired else list(batch)
        enc = tokenizer(texts, padding=True, truncation=True, max_length=max_length, **extra)
        out = {"input_ids": torch.as_tensor(enc["input_ids"], dtype=torch.long),
               "attention_mask": torch.as_tensor(enc["attention_mask"], dtype=torch.long)}
        if paired:
            out["text_ids"] = torch.as_tensor([b[0] for b in batch], dtype=torch.long)
        return out
    return collate


def mrr_at_k(run_ids: Sequence[Sequence], qrels: Dict, query_ids: Sequence, k: int = 10) -> float:
    """run_ids[i] = ranked document ids of query query_ids[i]; qrels[qid] = {doc id: relevance}."""
    total, n = 0.0, 0
    for qid, ranked in zip(query_ids, run_ids):
        rel = qrels.get(qid)
        if not rel:
            continue
        n += 1
        for r, did in enumerate(list(ranked)[:k]):
            if rel.get(did, 0) >= 1:
                total += 1.0 / (r + 1)
                break
    return round(total / max(n, 1), 5)


def recall_at_k(run_ids: Sequence[Sequence], qrels: Dict, query_ids: Sequence, k: int = 1000) -> float:
    total, n = 0.0, 0
    for qid, ranked in zip(query_ids, run_ids):
        rel = {d for d, s in qrels.get(qid, {}).items() if s >= 1}
        if not rel:
            continue
        n += 1
        total += len(rel.intersection(list(ranked)[:k])) / len(rel)
    return round(total / max(n, 1), 5)
