"""Deterministic synthetic inputs shared by the golden-vector generator, the
tests and bench.py.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Inputs are regenerated from ``numpy.random.default_rng(seed)`` everywhere, so
the golden fixtures under tests/golden/ only store *outputs* of the reference
(SURVEY.md §8c).  numpy is the same build in this container and on the GPU box.
"""
import numpy as np

K = 256
D = 768


def clustered_embeddings(seed: int, B: int, D: int = D, n_clusters: int = 64) -> np.ndarray:
    """Clustered-Gaussian 'document embeddings' [B, D] fp32."""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((n_clusters, D), dtype=np.float32)
    assign = rng.integers(0, n_clusters, size=B)
    noise = rng.standard_normal((B, D), dtype=np.float32)
    return (np.float32(0.7) * centers[assign] + np.float32(0.5) * noise).astype(np.float32)


def sample_centroids(seed: int, x: np.ndarray, M: int, K: int = K) -> np.ndarray:
    """Centroids [M, K, dsub]: K rows of x picked by a seeded permutation, sliced per
    sub-space (same construction as SURVEY.md §8d input A)."""
    B, D = x.shape
    assert B >= K and D % M == 0
    perm = np.random.default_rng(seed).permutation(B)[:K]
    return np.ascontiguousarray(
        x[perm].reshape(K, M, D // M).transpose(1, 0, 2)).astype(np.float32)


def gaussian(seed: int, shape) -> np.ndarray:
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def uniform_codes(seed: int, N: int, M: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, size=(N, M), dtype=np.uint8)


def golden_case_inputs(name: str, M: int, B: int, kind: str):
    """(seed, x [B,768], centroids [M,256,dsub]) of a tests/golden/quantize_<name>.npz fixture.
    kind: "sample" = K rows of x, "blend" = mean of two such samples, "gauss" = i.i.d. N(0,1)."""
    import zlib
    seed = zlib.crc32(name.encode()) & 0xFFFF
    if kind == "gauss":
        return seed, gaussian(seed, (B, D)), gaussian(seed + 1, (M, K, D // M))
    x = clustered_embeddings(seed, B)
    C = sample_centroids(seed + 1, x, M)
    if kind == "blend":
        C = (np.float32(0.5) * (C + sample_centroids(seed + 2, x, M))).astype(np.float32)
    return seed, x, C
