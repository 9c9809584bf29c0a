"""Reader / writer for the Faiss `IndexPQ` file the reference passes between pipeline steps
(`faiss.write_index(faiss.downcast_index(index.index), path)` at train/run_warmup.py:187; `faiss.read_index` at
evaluate/run_repconc_eval.py:42 and train/run_train_jpq.py:102) — SURVEY.md §8f row N3.

Layout restated from Faiss 1.7.x `impl/index_write.cpp` / `index_read.cpp` (little endian):

    fourcc   "IxPq"
    header   d:int32, ntotal:int64, dummy:int64 (1<<20), dummy:int64 (1<<20), is_trained:uint8, metric_type:int32
             [, metric_arg:float32 if metric_type > 1]
    pq       d:uint64, M:uint64, nbits:uint64, centroids: (n:uint64, float32[n])     n = M * 2^nbits * d/M
    codes    (n:uint64, uint8[n])                                                   n = ntotal * M
    search_type:int32, encode_signs:uint8, polysemous_ht:int32

No Faiss build is reachable from this environment, so the layout is UNVERIFIED against a file written by Faiss
itself; it is exercised by round-trip tests only.  metric_type 0 = METRIC_INNER_PRODUCT, 1 = METRIC_L2.
"""
from __future__ import annotations

import struct

import numpy as np
import torch

from .index import PQIndex

_DUMMY = 1 << 20


def write_index(index: PQIndex, path: str):
    pq = index.pq
    cent = index.pq.centroids.detach().cpu().numpy().astype("<f4").ravel()
    codes = index.codes.cpu().numpy().astype(np.uint8).ravel()
    with open(path, "wb") as f:
        f.write(b"IxPq")
        f.write(struct.pack("<iqqqBi", pq.d, index.ntotal, _DUMMY, _DUMMY, 1 if index.is_trained else 0,
                            int(index.metric_type)))
        f.write(struct.pack("<QQQ", pq.d, pq.M, pq.nbits))
        f.write(struct.pack("<Q", cent.size))
        f.write(cent.tobytes())
        f.write(struct.pack("<Q", codes.size))
        f.write(codes.tobytes())
        f.write(struct.pack("<iBi", 0, 0, 0))          # ST_PQ, encode_signs=false, polysemous_ht=0


def parse_index_file(path: str) -> dict:
    """Strict reader of the byte layout above (no device involved): d, M, ntotal, metric, is_trained, centroids
    [M,256,dsub] float32, codes [ntotal,M] uint8.  Anything that does not fit the layout — another fourcc, sizes that do
    not match d / M / ntotal, a short file, bytes after the last field — raises ValueError: the layout was restated from
    the published writer, not checked against a Faiss-written file, so a mismatch must be loud, never a silently wrong
    index."""
    def take(f, n, what):
        b = f.read(n)
        if len(b) != n:
            raise ValueError(f"truncated IndexPQ file: {what} needs {n} bytes, {len(b)} left")
        return b

    with open(path, "rb") as f:
        fourcc = f.read(4)
        if fourcc != b"IxPq":
            raise ValueError(f"not a Faiss IndexPQ file (fourcc {fourcc!r}, IxPq expected)")
        d, ntotal, _, _, trained, metric = struct.unpack("<iqqqBi", take(f, 4 + 8 * 3 + 1 + 4, "index header"))
        if metric > 1:
            take(f, 4, "metric_arg")
        if metric not in (0, 1) or trained not in (0, 1) or d <= 0 or ntotal < 0:
            raise ValueError(f"implausible IndexPQ header (d={d}, ntotal={ntotal}, is_trained={trained}, metric={metric})")
        d2, M, nbits = struct.unpack("<QQQ", take(f, 24, "ProductQuantizer header"))
        if d2 != d or nbits != 8 or M == 0 or d % M != 0:
            raise ValueError(f"unsupported ProductQuantizer (d={d2} vs index d={d}, M={M}, nbits={nbits})")
        (n,) = struct.unpack("<Q", take(f, 8, "centroid vector size"))
        if n != 256 * d:
            raise ValueError(f"centroid vector holds {n} floats, M*256*dsub = {256 * d} expected")
        cent = np.frombuffer(take(f, 4 * n, "centroids"), dtype="<f4")
        (n,) = struct.unpack("<Q", take(f, 8, "code vector size"))
        if n != ntotal * M:
            raise ValueError(f"code vector holds {n} bytes, ntotal*M = {ntotal * M} expected")
        codes = np.frombuffer(take(f, n, "codes"), dtype=np.uint8)
        search_type, encode_signs, polysemous_ht = struct.unpack("<iBi", take(f, 9, "search_type / encode_signs / polysemous_ht"))
        if search_type != 0:
            raise ValueError(f"IndexPQ search_type {search_type}: only ST_PQ (0) is supported")
        if f.read(1):
            raise ValueError("bytes after the last IndexPQ field: not the layout this reader knows")
    return {"d": d, "M": int(M), "ntotal": int(ntotal), "metric": metric, "is_trained": bool(trained),
            "centroids": cent.reshape(int(M), 256, d // int(M)).copy(), "codes": codes.reshape(int(ntotal), int(M)).copy()}


def read_index(path: str, device=None) -> PQIndex:
    p = parse_index_file(path)
    index = PQIndex(p["d"], p["M"], 8, p["metric"], device=device)
    index.set_centroids(torch.from_numpy(p["centroids"]))
    index.is_trained = p["is_trained"]
    if p["ntotal"]:
        index.add_codes(torch.from_numpy(p["codes"]))
    return index


# ---- the directory convention of the pipeline steps --------------------------------------------------------------
# evaluate/run_repconc_eval.py:39-43,57-58 and train/run_train_jpq.py:102-103: <dir>/index (the IndexPQ file) next to
# <dir>/corpus_ids.npy (index row -> corpus id, the order encode_corpus produced); train/run_warmup.py:187-189 writes both.
INDEX_FILE, CORPUS_IDS_FILE = "index", "corpus_ids.npy"


def save_index_dir(index: PQIndex, corpus_ids, out_dir: str):
    import os
    os.makedirs(out_dir, exist_ok=True)
    ids = np.asarray(corpus_ids)
    if len(ids) != index.ntotal:
        raise ValueError(f"{len(ids)} corpus ids for an index of {index.ntotal} rows")
    write_index(index, os.path.join(out_dir, INDEX_FILE))
    np.save(os.path.join(out_dir, CORPUS_IDS_FILE), ids)


def load_index_dir(in_dir: str, device=None):
    """-> (index, corpus_ids); raises FileNotFoundError when either file is missing (callers then encode the corpus)."""
    import os
    ip, cp = os.path.join(in_dir, INDEX_FILE), os.path.join(in_dir, CORPUS_IDS_FILE)
    if not (os.path.exists(ip) and os.path.exists(cp)):
        raise FileNotFoundError(f"{ip} / {cp}")
    index = read_index(ip, device=device)
    ids = np.load(cp)
    if len(ids) != index.ntotal:
        raise ValueError(f"{cp} holds {len(ids)} ids, the index {index.ntotal} rows")
    return index, ids
