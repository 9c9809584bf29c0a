"""PQ head of RepCONC behind the reference's own Python surface, computed by HIP kernels.

Mirror of models/repconc/modeling_repconc.py (same class, method and function names, argument
meaning, return layout, state_dict keys `rotation`, `centroids`, `dense_encoder.*`), so callers
such as RepCONCFinetuner / RepCONCEvaluater / JPQ can import it unchanged:

    from repconc_amd.models.repconc import RepCONC, sinkhorn_algorithm, decode, QuantizeOutput

What differs is where the arithmetic runs: quantize / decode / centring / Sinkhorn call
librepconc_hip.so (repconc_amd.ops); only the encoder forward and the rotation GEMM stay on
PyTorch-ROCm.  There is no CPU path: CPU tensors raise RepconcHipError.
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass
from typing import Optional, Union

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import Tensor, nn
from transformers.modeling_outputs import ModelOutput

from ... import _lib, ops
from ...sharded import SingleComm, TorchDistComm, assign_sinkhorn_sharded

logger = logging.getLogger(__name__)


@dataclass
class QuantizeOutput(ModelOutput):
    continuous_embeds: Optional[torch.FloatTensor] = None
    quantized_embeds: Optional[torch.FloatTensor] = None
    discrete_codes: Optional[torch.LongTensor] = None


def _dist_comm():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return TorchDistComm()
    return SingleComm()


class RepCONC(nn.Module):
    """modeling_repconc.py:28-134."""

    def __init__(self, config, dense_encoder, use_constraint: bool, sk_epsilon: Optional[float],
                 sk_iters: Optional[int]):
        super().__init__()
        self.config = config
        self.dense_encoder = dense_encoder
        D = dense_encoder.config.hidden_size
        M, K = config.MCQ_M, config.MCQ_K
        assert config.hidden_size % M == 0
        # K is fixed at 256 as in the reference (modeling_repconc.py:40); MCQ_M may be any divisor of hidden_size (:41): the
        # recipes' widths run on specialised kernels, the others on the run-time-width kernels with the same arithmetic
        if K != ops.K:
            raise _lib.RepconcHipError(f"MCQ_K={K} unsupported: the reference asserts MCQ_K == 256")
        # OPQ rotation (identity until the warm-up fills it) and the M x K sub-centroids
        self.register_buffer("rotation", torch.eye(D))
        self.centroids = nn.Parameter(torch.randn((M, K, config.hidden_size // M)))
        if getattr(config, "similarity_metric", None) == "METRIC_CENTROID_COS":
            self.normalize_centrodis()
        self.centroids.requires_grad = True
        self.use_constraint, self.sk_epsilon, self.sk_iters = use_constraint, sk_epsilon, sk_iters

    # ------------------------------------------------------------------ quantise / decode
    @torch.no_grad()
    def quantize(self, continuous_embeds: Tensor) -> Tensor:
        """codes int64 [B, M]: nearest centroid, or — with `use_constraint` — the argmax of the
        Sinkhorn transport plan that gives every centroid an equal share of the (global) batch.
        modeling_repconc.py:47-67."""
        if not self.use_constraint:
            return ops.assign_nearest(continuous_embeds, self.centroids, torch.int64)
        comm = _dist_comm()
        if comm.world == 1:
            codes, flags = ops.assign_sinkhorn(continuous_embeds, self.centroids, self.sk_epsilon, self.sk_iters)
        else:
            codes, flags = assign_sinkhorn_sharded(continuous_embeds, self.centroids, self.sk_epsilon,
                                                   self.sk_iters, comm)
        fl = int(flags.item())
        if fl & _lib.RC_FLAG_COMM:
            # a peer never arrived at an exchange (died, or skewed by more than the time-out): the codes are not the
            # batch's codes and the channel is out of step — stop here, as a process-group time-out would
            raise _lib.RepconcHipError("constrained assignment: the inter-rank exchange timed out (RC_FLAG_COMM); "
                                       "a peer rank is missing or more than RC_IPC_TIMEOUT_MS behind")
        if fl != 0:
            # the reference logs and returns (modeling_repconc.py:64-65); RC_FLAG_RANGE = sk_epsilon below ~3e-4, where
            # the reference's own exp(1/eps) has long overflowed fp64 (below 1.4e-3) and logs the same line
            logger.warning("Sinkhorn Algorithm returns nan/inf values.")
        return codes

    def decode(self, codes: Tensor) -> Tensor:
        return decode(codes, self.centroids)

    @staticmethod
    def center_distance_for_constraint(distances: Tensor) -> Tensor:
        """[M,B,K] fp32 -> (d - mid)/amp per sub-quantiser, the range taken over all ranks.
        modeling_repconc.py:73-85."""
        M = distances.shape[0]
        minmax = torch.cat([distances.amax(dim=(1, 2)), distances.amin(dim=(1, 2))]).float().contiguous()
        _dist_comm().allreduce_minmax_(minmax, M)
        mid = (minmax[:M] + minmax[M:]) / 2
        assert torch.all(minmax[:M] - mid + 1e-5 > 0)
        return ops.centre_(distances.float().contiguous().clone(), minmax)

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids, attention_mask, discrete_codes=None, return_code=False,
                return_quantized_embedding=False) -> QuantizeOutput:
        """encoder -> rotation -> (COS: per-sub-vector L2 normalise) -> optional quantise/decode.
        modeling_repconc.py:87-110."""
        dense = self.dense_encoder(input_ids=input_ids, attention_mask=attention_mask)
        rotated = dense @ self.rotation.T
        if getattr(self.config, "similarity_metric", None) == "METRIC_CENTROID_COS":
            B = rotated.shape[0]
            rotated = F.normalize(rotated.reshape(B, self.config.MCQ_M, -1), p=2, dim=-1).reshape(B, -1)
        if discrete_codes is None and (return_code or return_quantized_embedding):
            discrete_codes = self.quantize(rotated)
        quantized = self.decode(discrete_codes) if return_quantized_embedding else None
        return QuantizeOutput(continuous_embeds=rotated, quantized_embeds=quantized, discrete_codes=discrete_codes)

    @torch.no_grad()
    def normalize_centrodis(self):  # (sic) the reference's spelling, modeling_repconc.py:112-116
        data = self.centroids.data
        if data.is_cuda:
            ops.normalize_centroids_(data)
        else:  # parameters still on the host (constructor): round-trip through the device kernel
            tmp = data.to(torch.device("cuda", torch.cuda.current_device())).contiguous()
            data.copy_(ops.normalize_centroids_(tmp).cpu())

    # ------------------------------------------------------------------ persistence
    def save_pretrained(self, output_dir: str):
        """pytorch_model.bin (rotation, centroids, dense_encoder.*) + config + dense_encoder/.
        modeling_repconc.py:118-122."""
        os.makedirs(output_dir, exist_ok=True)
        torch.save(self.state_dict(), os.path.join(output_dir, "pytorch_model.bin"))
        self.config.save_pretrained(output_dir)
        self.dense_encoder.save_pretrained(os.path.join(output_dir, "dense_encoder"))

    @classmethod
    def from_pretrained(cls, load_dir: str, use_constraint, sk_epsilon, sk_iters):
        from ..dense import AutoDense
        enc = AutoDense.from_pretrained(os.path.join(load_dir, "dense_encoder"))
        model = cls(enc.config, enc, use_constraint=use_constraint, sk_epsilon=sk_epsilon, sk_iters=sk_iters)
        state = torch.load(os.path.join(load_dir, "pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(state)
        return model


@torch.no_grad()
def sinkhorn_algorithm(out: Tensor, epsilon: float, sinkhorn_iterations: int, use_distrib_train: bool) -> Tensor:
    """Transport plan Q [M,K,B] fp64 whose columns sum to 1 — modeling_repconc.py:137-165.

    The plan is rebuilt from the row potentials f after `sinkhorn_iterations` iterations:
    Q[:,:,b] = softmax_k(out/eps + f).  A tensor of fp32-representable values (what RepCONC.quantize passes:
    `-centred.double().transpose(1,2)`) runs on the streaming sweep over the fp32 table; any other fp64
    tensor on the general fp64 kernels (rc_sk64_rows / rc_sk64_cols) — same potentials, the caller's data."""
    if out.dim() != 3:
        raise ValueError("out must be [M, K, B]")
    distributed = bool(use_distrib_train and dist.get_world_size() > 1)
    d = (-out).transpose(1, 2).contiguous().float()
    exact32 = torch.equal(d.double(), (-out).transpose(1, 2)) if d.numel() else True
    if distributed:       # every rank must take the same route
        t = torch.tensor([0.0 if exact32 else 1.0], dtype=torch.float64, device=out.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exact32 = float(t.item()) == 0.0
    if not exact32:
        o64 = out.double().contiguous()
        f = ops.sinkhorn_potentials_f64(o64, epsilon, sinkhorn_iterations, TorchDistComm().allgather if distributed else None)
        return torch.softmax(o64 / epsilon + f[:, :, None], dim=1)
    # the sweeps take a centred table (|d| <= 1, as center_distance_for_constraint produces): a wider one is rescaled
    # together with eps by a power of two — L = d/eps is unchanged, bit for bit
    amax = float(d.abs().max()) if d.numel() else 0.0
    eps_k = epsilon
    if use_distrib_train and dist.get_world_size() > 1:
        t = torch.tensor([amax], dtype=torch.float64, device=d.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        amax = float(t.item())
    if amax > 1.0:
        import math
        sc = 2.0 ** math.ceil(math.log2(amax))
        d, eps_k = d / sc, epsilon / sc
    comm = TorchDistComm() if (use_distrib_train and dist.get_world_size() > 1) else SingleComm()
    st = ops.SinkhornState(d)
    rows = st.sweep(eps_k, 0, None)
    for t in range(1, sinkhorn_iterations):
        rows = st.sweep(eps_k, t, comm.allgather(rows))
    f = st.potentials(sinkhorn_iterations, comm.allgather(rows))
    return torch.softmax(out / epsilon + f[:, :, None], dim=1)


def decode(codes: Union[np.ndarray, Tensor], centroids: Union[np.ndarray, Tensor]):
    """codes [n, M] -> concatenated centroids [n, D]; torch (differentiable w.r.t. centroids) or
    numpy in / numpy out.  modeling_repconc.py:168-184.  Both variants run the HIP gather."""
    if isinstance(codes, torch.Tensor):
        assert isinstance(centroids, torch.Tensor)
        return ops.decode(codes, centroids)
    if isinstance(codes, np.ndarray):
        dev = torch.device("cuda", torch.cuda.current_device())
        c = centroids.detach() if isinstance(centroids, torch.Tensor) else torch.from_numpy(np.asarray(centroids))
        ct = torch.from_numpy(np.ascontiguousarray(codes))
        if ct.dtype != torch.uint8:
            ct = ct.to(torch.int64)
        return ops.decode_raw(ct.to(dev), c.to(dev)).cpu().numpy()
    raise NotImplementedError()
