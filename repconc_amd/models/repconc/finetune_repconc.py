"""Stage-1 (constrained clustering) fine-tuning with the reference's names — models/repconc/finetune_repconc.py:
`DataTrainingArguments`, `RepCONCFinetuneArguments`, `FinetuneCollator`, `QDRelDataset`, `RepCONC_Norm_Centroid_Callback`,
`RepCONCFinetuner`, `eval_balance`, `test_quantize` — on the transformers 5.x Trainer and this package's `GradCache`.

What is kept of the reference: the two-pass cached-gradient step (encode every tower without graph
in `cache_chunk_size` chunks -> constrained `quantize` of cat(pos, neg) over the GLOBAL batch -> contrastive loss on
(query, decode(codes)) with false-negative / duplicate masks and dynamic hard negatives -> second pass per chunk with
the cached gradient applied to the continuous embedding (straight-through) and to decode(codes) (-> centroids) plus the
weighted MSE), the cross-rank gather of representations and ids, the three optimiser groups, centroid re-normalisation
for METRIC_CENTROID_COS.  What changed: no use of Trainer members that 5.x dropped (`use_amp`, `scaler`, `use_apex`,
`sharded_ddp`, `_prepare_inputs` on nested dicts), fp16 scaling goes through the accelerator's scaler.
The arithmetic of quantize / decode / decode-backward / the balance statistics is `repconc_amd.ops`.
"""
from __future__ import annotations

import inspect
import json
import logging
import os
import random
from collections import defaultdict
from contextlib import nullcontext
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import Tensor, nn
from torch.utils.data import Dataset
from transformers import Trainer, TrainerCallback, TrainingArguments

from ...diagnostics import eval_balance, test_quantize  # noqa: F401  (same names as finetune_repconc.py:580-613)
from ...gradcache import GradCache
from .modeling_repconc import RepCONC

logger = logging.getLogger(__name__)


@dataclass
class DataTrainingArguments:
    qrel_path: str = field()
    query_path: str = field()
    corpus_path: str = field()
    valid_qrel_path: str = field()
    valid_query_path: str = field()
    valid_corpus_path: str = field()
    max_query_len: int = field()
    max_doc_len: int = field()


@dataclass
class RepCONCFinetuneArguments(TrainingArguments):
    """finetune_repconc.py:44-58."""
    negative_per_query: int = field(default=1)
    dynamic_topk_hard_negative: int = field(default=None)
    centroid_learning_rate: float = field(default=1e-3)
    temperature: float = field(default=1.0)
    mse_loss_weight: float = field(default=0)
    not_use_constraint: bool = field(default=False)
    negative: str = field(default="random", metadata={"help": "inbatch, random, or the path of a qid -> [docid] json"})
    cache_chunk_size: int = field(default=-1)
    seed: int = field(default=2022)
    remove_unused_columns: Optional[bool] = field(default=False)


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


class FinetuneCollator:
    """Tokenises queries, positives and (flattened) negatives of a batch; ids travel along so that the trainer can mask
    in-batch false negatives after the cross-rank gather.  finetune_repconc.py:61-126."""

    def __init__(self, tokenizer, max_query_len: int, max_doc_len: int):
        self.tokenizer, self.max_query_len, self.max_doc_len = tokenizer, max_query_len, max_doc_len
        try:
            typed = "input_text_type" in inspect.signature(tokenizer.__call__).parameters
        except (TypeError, ValueError):
            typed = False
        self._q = {"input_text_type": "query"} if typed else {}
        self._d = {"input_text_type": "doc"} if typed else {}

    def _tok(self, texts, max_len, extra):
        return self.tokenizer(texts, padding=True, return_tensors="pt", add_special_tokens=True, return_attention_mask=True,
                              return_token_type_ids=False, truncation=True, max_length=max_len, **extra)

    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        batch = {"query_input": self._tok([f["query"] for f in features], self.max_query_len, self._q),
                 "pos_doc_input": self._tok([f["pos_doc"] for f in features], self.max_doc_len, self._d),
                 "qids": torch.tensor([f["qid"] for f in features], dtype=torch.long),
                 "pos_docids": torch.tensor([f["pos_docid"] for f in features], dtype=torch.long)}
        if "neg_docs" in features[0]:
            batch["neg_doc_input"] = self._tok([t for f in features for t in f["neg_docs"]], self.max_doc_len, self._d)
            batch["neg_docids"] = torch.tensor([d for f in features for d in f["neg_docids"]], dtype=torch.long)
        return batch


class QDRelDataset(Dataset):
    """(query, one random positive, `negative_per_query` negatives) per item, texts tokenised later by the collator.
    Queries / documents are addressed by line offset; `negative` = "inbatch" | "random" | path of a json with
    qid -> [docid] hard negatives.  finetune_repconc.py:129-218."""

    def __init__(self, tokenizer, qrel_path, query_path, corpus_path, max_query_len, max_doc_len, negative,
                 negative_per_query, rel_threshold=1, verbose=True):
        self.tokenizer = tokenizer
        self.queries, q_off = [], {}
        with open(query_path) as f:
            for i, line in enumerate(f):
                qid, text = line.split("\t")
                q_off[qid] = i
                self.queries.append(text.strip())
        self.corpus, d_off = [], {}
        with open(corpus_path) as f:
            for i, line in enumerate(f):
                parts = line.strip().split("\t")
                d_off[parts[0]] = i
                self.corpus.append(str(tokenizer.sep_token).join(p.strip() for p in parts[1:]).strip()[:10000])
        qrels = defaultdict(list)
        with open(qrel_path) as f:
            for line in f:
                qid, _, docid, rel = line.split()
                if int(rel) >= rel_threshold:
                    qrels[q_off[qid]].append(d_off[docid])
        self.qrels = dict(qrels)
        self.qids = sorted(self.qrels)
        self.negative_per_query = negative_per_query
        if negative in ("inbatch", "random"):
            self.negative = negative
        else:
            with open(negative) as f:
                self.negative = {q_off[q]: [d_off[d] for d in docs] for q, docs in json.load(f).items()}
        self.max_query_len, self.max_doc_len = max_query_len, max_doc_len

    def get_qrels(self):
        return self.qrels

    def __len__(self):
        return len(self.qids)

    def __getitem__(self, index):
        qid = self.qids[index]
        pos = random.choice(self.qrels[qid])
        item = {"query": self.queries[qid], "pos_doc": self.corpus[pos], "pos_docid": pos, "qid": qid}
        if self.negative == "inbatch":
            assert self.negative_per_query == 0
            return item
        pool = range(len(self.corpus)) if self.negative == "random" else self.negative[qid]
        negs = random.sample(pool, self.negative_per_query)
        item.update(neg_docids=negs, neg_docs=[self.corpus[d] for d in negs])
        return item


class RepCONC_Norm_Centroid_Callback(TrainerCallback):
    def on_step_end(self, args, state, control, model=None, **kwargs):
        _unwrap(model).normalize_centrodis()


class RepCONCFinetuner(Trainer):
    """`RepCONCFinetuner(qrels, model=..., args=RepCONCFinetuneArguments, train_dataset=..., data_collator=...)`."""

    def __init__(self, qrels, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.qrels = qrels
        self._gc_scaler = getattr(getattr(self, "accelerator", None), "scaler", None) if self.args.fp16 else None
        if self.args.cache_chunk_size != -1:
            self.gc = GradCache(models=[self.model], chunk_sizes=self.args.cache_chunk_size,
                                loss_fn=self.compute_contrastive_loss, get_rep_fn=lambda out: out.continuous_embeds,
                                fp16=bool(self.args.fp16 and self._gc_scaler is not None), scaler=self._gc_scaler)
        if getattr(self.model.config, "similarity_metric", None) == "METRIC_CENTROID_COS":
            self.add_callback(RepCONC_Norm_Centroid_Callback)

    # ------------------------------------------------------------------ helpers
    def _world(self):
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _rank(self):
        return dist.get_rank() if self._world() > 1 else 0

    def _to_device(self, td):
        dev = self.args.device
        return {k: v.to(dev, non_blocking=True) for k, v in dict(td).items() if isinstance(v, torch.Tensor)}

    def split_tensor_dict(self, td: Dict[str, Tensor]):
        keys = list(td)
        parts = [td[k].split(self.args.cache_chunk_size) for k in keys]
        return [dict(zip(keys, p)) for p in zip(*parts)]

    def gather_tensors(self, *tt: Tensor):
        """Every rank's rows in rank order, the local rows being the live tensors.  finetune_repconc.py:453-462."""
        if self._world() == 1:
            return list(tt)
        out = []
        for t in tt:
            parts = [torch.empty_like(t) for _ in range(self._world())]
            dist.all_gather(parts, t.contiguous())
            parts[self._rank()] = t
            out.append(torch.cat(parts))
        return out

    def _autocast(self):
        if self.args.bf16:
            return torch.autocast("cuda", dtype=torch.bfloat16)
        if self.args.fp16:
            return torch.autocast("cuda", dtype=torch.float16)
        return nullcontext()

    # ------------------------------------------------------------------ one step
    def training_step(self, model, inputs, num_items_in_batch=None) -> torch.Tensor:
        """finetune_repconc.py:245-281: returns the (detached) contrastive loss; the backward passes happen inside."""
        if self.args.cache_chunk_size == -1:
            raise NotImplementedError("RepCONC stage 1 trains with cached gradients: set --cache_chunk_size")
        if self.args.gradient_accumulation_steps > 1:
            raise ValueError("gradient accumulation is what the gradient cache replaces")
        model.train()
        core = _unwrap(model)
        towers = [("query", self.split_tensor_dict(self._to_device(inputs["query_input"])), None),
                  ("pos", self.split_tensor_dict(self._to_device(inputs["pos_doc_input"])), inputs["pos_docids"])]
        if "neg_doc_input" in inputs:
            towers.append(("neg", self.split_tensor_dict(self._to_device(inputs["neg_doc_input"])), inputs["neg_docids"]))
        reps, rnd = {}, {}
        for name, chunks, _ in towers:
            reps[name], rnd[name] = self.gc.forward_no_grad(model, chunks)
        # constrained clustering over the (global) batch of documents — the hot path (:296-318)
        doc_names = [n for n, _, _ in towers if n != "query"]
        docs = torch.vstack([reps[n] for n in doc_names])
        codes_all = core.quantize(docs)
        codes, quant, off = {}, {}, 0
        for n in doc_names:
            codes[n] = codes_all[off:off + len(reps[n])]
            quant[n] = core.decode(codes[n])
            off += len(reps[n])
        if self.state.global_step % max(int(self.args.logging_steps), 1) == 0:
            self.log(test_quantize(docs, core, self._rank() if self._world() > 1 else -1, block_id=0))
        dev = self.args.device
        qids = inputs["qids"].to(dev).contiguous()
        ids = {n: t.to(dev).contiguous() for n, _, t in towers if t is not None}
        g_query, = self.gather_tensors(reps["query"].contiguous())
        g_quant = {n: self.gather_tensors(quant[n].contiguous())[0] for n in doc_names}
        g_qids, = self.gather_tensors(qids)
        g_ids = {n: self.gather_tensors(ids[n])[0] for n in doc_names}
        all_docs = torch.vstack([g_quant[n] for n in doc_names])
        all_ids = torch.hstack([g_ids[n] for n in doc_names])
        (g_q, g_d), loss = self.gc.build_cache(g_query, all_docs, qids=g_qids, docids=all_ids)
        grads, off = {"query": g_q}, 0
        for n in doc_names:
            grads[n] = g_d[off:off + len(g_quant[n])]
            off += len(g_quant[n])
        self._forward_backward(model, [(chunks, grads[name], rnd[name], codes.get(name)) for name, chunks, _ in towers])
        return loss.detach().float()

    def _forward_backward(self, model, plan):
        """Second pass with graph, chunk by chunk (finetune_repconc.py:346-396).  `plan`: [(chunks, cached grads over the
        GLOBAL rows of that tower, RNG snapshots, local codes | None for the query tower)]."""
        W, r, cs = self._world(), self._rank(), self.args.cache_chunk_size
        scaler = self._gc_scaler
        for ti, (chunks, cache, rnd, codes) in enumerate(plan):
            n_local = sum(next(iter(c.values())).shape[0] for c in chunks)
            assert W * n_local == len(cache), f"{W} * {n_local} != {len(cache)}"
            base = r * n_local
            for ci, chunk in enumerate(chunks):
                lo = ci * cs
                with rnd[ci], self._autocast():
                    if codes is None:
                        out = model(**chunk)
                    else:
                        out = model(discrete_codes=codes[lo:lo + cs], return_quantized_embedding=True, **chunk)
                    n = out.continuous_embeds.shape[0]
                    g = cache[base + lo: base + lo + n].flatten()
                    obj = torch.dot(g, out.continuous_embeds.flatten().to(g.dtype))
                    if codes is not None:
                        obj = obj + torch.dot(g, out.quantized_embeds.flatten().to(g.dtype))
                        mse = ((out.quantized_embeds - out.continuous_embeds) ** 2).sum(-1).mean() * self.args.mse_loss_weight
                        obj = obj + (scaler.scale(mse) if scaler is not None else mse)
                last = ci + 1 == len(chunks) and ti + 1 == len(plan)
                sync = nullcontext() if (W == 1 or last or not hasattr(model, "no_sync")) else model.no_sync()
                with sync:
                    obj.backward()

    # ------------------------------------------------------------------ loss
    def compute_contrastive_loss(self, query_embeds, doc_embeds, qids, docids):
        """In-batch softmax over every gathered document, labels on the diagonal.  finetune_repconc.py:398-431."""
        nq = query_embeds.shape[0]
        labels = torch.arange(nq, dtype=torch.long, device=query_embeds.device)
        mask = torch.logical_or(self._compute_mask_for_false_negative(qids, docids),
                                self._compute_mask_for_duplicate_negative(qids, docids)).float()
        sim = query_embeds @ doc_embeds.T
        if getattr(self.model.config, "similarity_metric", None) == "METRIC_CENTROID_COS":
            sim = sim / self.model.config.MCQ_M
        if self.args.temperature != 1:
            sim = sim / self.args.temperature
        sim = sim - 10000.0 * mask
        topk = self.args.dynamic_topk_hard_negative
        if topk is not None and topk > 0:
            drop = torch.ones_like(sim)
            neg = sim.detach().clone()
            neg.scatter_(1, labels[:, None], -10000.0)
            drop.scatter_(1, torch.topk(neg, topk).indices, 0)
            drop.scatter_(1, labels[:, None], 0)
            sim = sim - 10000.0 * drop
        return F.cross_entropy(sim, labels)

    @torch.no_grad()
    def _compute_mask_for_false_negative(self, qids, docids):
        mask = torch.zeros((len(qids), len(docids)), dtype=torch.bool, device=qids.device)
        for i, qid in enumerate(qids.tolist()):
            for d in self.qrels.get(qid, ()):
                mask[i] |= docids == d
        mask.fill_diagonal_(False)
        return mask

    @torch.no_grad()
    def _compute_mask_for_duplicate_negative(self, qids, docids):
        dup = torch.triu(docids[:, None] == docids[None, :], diagonal=1).any(dim=0, keepdim=True).repeat(len(qids), 1)
        dup.fill_diagonal_(False)
        return dup

    # ------------------------------------------------------------------ Trainer plumbing
    def floating_point_ops(self, inputs):
        return 0

    def _save(self, output_dir: Optional[str] = None, state_dict=None):
        output_dir = output_dir or self.args.output_dir
        _unwrap(self.model).save_pretrained(output_dir)          # pytorch_model.bin + config + dense_encoder/ (:466-469)

    def create_optimizer(self, *args, **kwargs):
        """Three groups: decayed / undecayed encoder parameters, centroids at `centroid_learning_rate` without decay.
        finetune_repconc.py:476-528."""
        if self.optimizer is None:
            core = _unwrap(self.model)
            no_decay = {n for n, p in core.named_parameters() if p.ndim < 2 or "bias" in n or "LayerNorm" in n or "layer_norm" in n}
            named = [(n, p) for n, p in core.named_parameters() if p.requires_grad]
            groups = [
                {"params": [p for n, p in named if n not in no_decay and "centroids" not in n], "weight_decay": self.args.weight_decay},
                {"params": [p for n, p in named if n in no_decay and "centroids" not in n], "weight_decay": 0.0},
                {"params": [p for n, p in named if "centroids" in n], "weight_decay": 0.0, "lr": self.args.centroid_learning_rate},
            ]
            logger.info("optimizer groups: %s", [len(g["params"]) for g in groups])
            self.optimizer = torch.optim.AdamW(groups, lr=self.args.learning_rate, betas=(self.args.adam_beta1, self.args.adam_beta2),
                                               eps=self.args.adam_epsilon)
        return self.optimizer
