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


def read_index(path: str, device=None) -> PQIndex:
    with open(path, "rb") as f:
        if f.read(4) != b"IxPq":
            raise ValueError("not a Faiss IndexPQ file (fourcc IxPq expected)")
        d, ntotal, _, _, trained, metric = struct.unpack("<iqqqBi", f.read(4 + 8 * 3 + 1 + 4))
        if metric > 1:
            f.read(4)
        d2, M, nbits = struct.unpack("<QQQ", f.read(24))
        if d2 != d or nbits != 8:
            raise ValueError(f"unsupported ProductQuantizer (d={d2}, nbits={nbits})")
        (n,) = struct.unpack("<Q", f.read(8))
        cent = np.frombuffer(f.read(4 * n), dtype="<f4")
        (n,) = struct.unpack("<Q", f.read(8))
        codes = np.frombuffer(f.read(n), dtype=np.uint8)
    index = PQIndex(d, M, 8, metric, device=device)
    index.set_centroids(torch.from_numpy(cent.reshape(M, 256, d // M).copy()))
    index.is_trained = bool(trained)
    if ntotal:
        index.add_codes(torch.from_numpy(codes.reshape(ntotal, M).copy()))
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
