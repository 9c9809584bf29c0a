"""Gradient caching for contrastive training with batches larger than one forward pass can hold.

The reference trains stage 1 through the third-party `grad_cache` package (unpinned git HEAD, setup.py:22; used at
models/repconc/finetune_repconc.py:26,234-241,283-284,302-303,312-314,340-341,357), which is not installable offline.
This module provides the part of its surface the reference touches, written from the published algorithm
(Gao et al., "Scaling Deep Contrastive Learning Batch Size under Memory Limited Setup", 2021):

    gc = GradCache(models=[model], chunk_sizes=64, loss_fn=f, get_rep_fn=lambda out: out.continuous_embeds,
                   fp16=False, scaler=None)
    reps, rnd_states = gc.forward_no_grad(model, chunked_inputs)      # representations without graph + RNG snapshots
    grads, loss = gc.build_cache(*reps_of_every_tower, **loss_kwargs)  # d loss / d reps (the "cache"), detached loss
    with rnd_states[i]: out = model(**chunked_inputs[i])               # second pass replays the dropout masks

`RandContext` snapshots the CPU and current-device RNG in front of a chunk's first forward; entered as a context
manager it restores that state for the second forward and puts the ambient state back on exit.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Union

import torch


class RandContext:
    def __init__(self, *tensors):
        self.device = next((t.device for t in tensors if isinstance(t, torch.Tensor) and t.is_cuda), None)
        self.cpu_state = torch.get_rng_state()
        self.cuda_state = torch.cuda.get_rng_state(self.device) if self.device is not None else None
        self._saved = None

    def __enter__(self):
        self._saved = (torch.get_rng_state(),
                       torch.cuda.get_rng_state(self.device) if self.device is not None else None)
        torch.set_rng_state(self.cpu_state)
        if self.device is not None:
            torch.cuda.set_rng_state(self.cuda_state, self.device)
        return self

    def __exit__(self, *exc):
        cpu, cuda = self._saved
        torch.set_rng_state(cpu)
        if cuda is not None:
            torch.cuda.set_rng_state(cuda, self.device)
        self._saved = None


class GradCache:
    def __init__(self, models: Sequence[torch.nn.Module], chunk_sizes: Union[int, Sequence[int]], loss_fn: Callable,
                 split_input_fn: Callable = None, get_rep_fn: Callable = None, fp16: bool = False, scaler=None):
        self.models = list(models)
        self.chunk_sizes = [chunk_sizes] * len(self.models) if isinstance(chunk_sizes, int) else list(chunk_sizes)
        self.loss_fn = loss_fn
        self.split_input_fn = split_input_fn
        self.get_rep_fn = get_rep_fn
        self.fp16 = fp16
        self.scaler = scaler
        if fp16 and scaler is None:
            raise ValueError("fp16 gradient caching needs a GradScaler")

    # -- first pass --------------------------------------------------------------------------------------------
    def get_reps(self, model_out) -> torch.Tensor:
        return model_out if self.get_rep_fn is None else self.get_rep_fn(model_out)

    def model_call(self, model, model_input):
        if isinstance(model_input, dict):
            return model(**model_input)
        if isinstance(model_input, (list, tuple)):
            return model(*model_input)
        return model(model_input)

    @torch.no_grad()
    def forward_no_grad(self, model, model_inputs: List):
        """model_inputs: the already chunked inputs of ONE tower.  -> (reps [n, D] without graph, [RandContext])."""
        reps, states = [], []
        for chunk in model_inputs:
            tensors = list(chunk.values()) if isinstance(chunk, dict) else list(chunk) if isinstance(chunk, (list, tuple)) else [chunk]
            states.append(RandContext(*tensors))
            with torch.autocast("cuda", enabled=self.fp16):
                reps.append(self.get_reps(self.model_call(model, chunk)))
        return torch.cat(reps, 0), states

    # -- the cache ---------------------------------------------------------------------------------------------
    def compute_loss(self, *reps, **loss_kwargs):
        return self.loss_fn(*reps, **loss_kwargs)

    def build_cache(self, *reps: torch.Tensor, **loss_kwargs):
        """Loss on the full-batch representations; returns ([d loss / d rep for every tower], loss.detach()).
        With fp16 the loss is scaled by the GradScaler before the backward, so the cached gradients carry the scale."""
        leaves = [r.detach().requires_grad_() for r in reps]
        with torch.autocast("cuda", enabled=self.fp16):
            loss = self.compute_loss(*leaves, **loss_kwargs)
        (self.scaler.scale(loss) if self.fp16 else loss).backward()
        return [leaf.grad for leaf in leaves], loss.detach()
