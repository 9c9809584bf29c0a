"""ctypes wrapper of oracle/pq_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pq_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_quantize.restype = C.c_int
        _lib.orc_sinkhorn_codes.restype = C.c_int
    return _lib


def _f(a):
    return a.ctypes.data_as(C.c_void_p)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n: int):
    """OpenMP thread count of the C restatement (the reference's eval default is ONE Faiss thread, evaluate_repconc.py:37)."""
    lib().orc_set_num_threads(int(n))


def quantize(x, centroids, use_constraint, eps=0.003, iters=100):
    x = np.ascontiguousarray(x, np.float32)
    Cn = np.ascontiguousarray(centroids, np.float32)
    B, D = x.shape
    M, K, dsub = Cn.shape
    assert K == 256 and D == M * dsub
    codes = np.empty((B, M), np.uint8)
    ws = np.empty((M, B, K), np.float32)
    flags = lib().orc_quantize(_f(x), C.c_int64(D), _f(Cn), C.c_int64(B), M, dsub, int(use_constraint),
                               C.c_double(eps), int(iters), _f(codes), _f(ws))
    return codes, flags


def dist_table(x, centroids):
    x = np.ascontiguousarray(x, np.float32)
    Cn = np.ascontiguousarray(centroids, np.float32)
    B, D = x.shape
    M, K, dsub = Cn.shape
    d = np.empty((M, B, K), np.float32)
    lib().orc_dist_table(_f(x), C.c_int64(D), _f(Cn), C.c_int64(B), M, dsub, _f(d))
    return d


def minmax(d):
    M, B, _ = d.shape
    mm = np.empty(2 * M, np.float32)
    lib().orc_minmax(_f(d), C.c_int64(B), M, _f(mm))
    return mm


def centre_(d, mm):
    M, B, _ = d.shape
    lib().orc_centre(_f(d), _f(np.ascontiguousarray(mm, np.float32)), C.c_int64(B), M)
    return d


def decode(codes, centroids):
    codes = np.ascontiguousarray(codes, np.uint8)
    Cn = np.ascontiguousarray(centroids, np.float32)
    n, M = codes.shape
    dsub = Cn.shape[2]
    out = np.empty((n, M * dsub), np.float32)
    lib().orc_decode(_f(codes), _f(Cn), C.c_int64(n), M, dsub, _f(out))
    return out


def adc_search(codes, centroids, q, k, tile=None):
    """tile=None: one query per thread, each streaming the whole code array (Faiss's IndexPQ search loop); tile=rows: the
    cache-blocked form (orc_adc_search_tiled; 0 = its default tile) — identical results."""
    codes = np.ascontiguousarray(codes, np.uint8)
    Cn = np.ascontiguousarray(centroids, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    N, M = codes.shape
    nq = q.shape[0]
    scores = np.empty((nq, k), np.float32)
    ids = np.empty((nq, k), np.int64)
    if tile is None:
        lib().orc_adc_search(_f(codes), C.c_int64(N), M, Cn.shape[2], _f(Cn), _f(q), nq, k, _f(scores), _f(ids))
    else:
        lib().orc_adc_search_tiled(_f(codes), C.c_int64(N), M, Cn.shape[2], _f(Cn), _f(q), nq, k, C.c_int64(int(tile)),
                                   _f(scores), _f(ids))
    return scores, ids


def first_touch_copy(a):
    """A copy of `a` whose pages were first written by all OpenMP threads in parallel (NUMA placement for the CPU baselines)."""
    a = np.ascontiguousarray(a)
    out = np.empty_like(a)
    lib().orc_first_touch_copy(_f(out), _f(a), C.c_int64(a.nbytes))
    return out
