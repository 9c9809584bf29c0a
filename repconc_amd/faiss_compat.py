"""The handful of `faiss.*` idioms the reference's call sites use on the index object, for `PQIndex`.

A maintainer who swaps `import faiss` for `from repconc_amd import faiss_compat as faiss` in
evaluate/run_repconc_eval.py, models/jpq/finetune_jpq.py and train/run_warmup.py keeps these lines unchanged:

    faiss.copy_array_to_vector(centroids.ravel(), index.pq.centroids)   evaluate_repconc.py:84-85, run_repconc_eval.py:123-127,
                                                                        finetune_jpq.py:211-213
    faiss.vector_to_array(index.pq.centroids) / (index.codes)           run_warmup.py:124-125, finetune_jpq.py:161
    faiss.IndexPQ(D, M, 8, faiss.METRIC_INNER_PRODUCT)                  evaluate_repconc.py:81
    faiss.write_index(index, path) / faiss.read_index(path)             run_warmup.py:187, run_repconc_eval.py:42
    faiss.omp_set_num_threads(n)                                        run_repconc_eval.py:149 (no-op: the scan runs on the GPU)

The "vectors" are torch tensors here: `copy_array_to_vector` writes in place (the index's centroid table is ONE
resident tensor shared by `index.pq.centroids` and the search), `vector_to_array` returns a numpy copy.
"""
from __future__ import annotations

import numpy as np
import torch

from .faiss_io import read_index, write_index  # noqa: F401  (re-exported)
from .index import METRIC_INNER_PRODUCT, PQIndex

METRIC_L2 = 1


def IndexPQ(d: int, M: int, nbits: int = 8, metric=METRIC_INNER_PRODUCT) -> PQIndex:
    return PQIndex(d, M, nbits, metric)


def copy_array_to_vector(array, vector: torch.Tensor) -> None:
    """`vector` is `index.pq.centroids` (or any tensor of the same size): overwritten in place, [m][k][j] order."""
    src = array.detach() if isinstance(array, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(array))
    if src.numel() != vector.numel():
        raise ValueError(f"size mismatch: {src.numel()} values for a vector of {vector.numel()}")
    vector.copy_(src.reshape(vector.shape).to(vector.device, vector.dtype))


def vector_to_array(vector) -> np.ndarray:
    """Flat numpy copy of `index.pq.centroids` (float32) or `index.codes` (uint8, row-major [ntotal * M])."""
    # tensors, and the fan-out view a multi-device index exposes as pq.centroids (multi_index._FanoutCentroids)
    t = vector.detach() if hasattr(vector, "detach") else torch.as_tensor(vector)
    return t.reshape(-1).cpu().numpy().copy()


def omp_set_num_threads(n: int) -> None:
    """Accepted and ignored: there is no OpenMP pool behind the search."""


def downcast_index(index):
    return index
