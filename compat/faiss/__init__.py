"""`import faiss` for the reference's SCRIPTS when Faiss itself is not installed — opt in by putting `compat/` on the path
(next to `compat/repconc`, which serves the `repconc.*` imports):

    PYTHONPATH=/path/to/repo/compat:/path/to/repo  python /path/to/reference/src/repconc/evaluate/run_repconc_eval.py ...

The reference's own modules that drive Faiss objects in depth (models/repconc/evaluate_repconc.py, train/run_warmup.py,
models/jpq/finetune_jpq.py) are replaced as a whole by their `repconc_amd` counterparts; what remains are the handful of
`faiss.*` idioms in the entry scripts, all on the index object:

    faiss.read_index / faiss.write_index            evaluate/run_repconc_eval.py:42,57, train/run_train_jpq.py:102
    faiss.copy_array_to_vector(c, index.pq.centroids)   evaluate/run_repconc_eval.py:126
    faiss.omp_set_num_threads(n)                    evaluate/run_repconc_eval.py:149
    faiss.IndexPQ / faiss.IndexIVFPQ                type annotations (run_repconc_eval.py:123, finetune_jpq.py:145)
    import faiss.contrib.torch_utils                finetune_jpq.py:9 (tensor in / tensor out is what PQIndex.search does)

This is `repconc_amd.faiss_compat` under Faiss's name; anything else (`index_factory`, `StandardGpuResources`, …) is not
provided and raises AttributeError — those call sites live in the modules listed above.  A real Faiss installation, if
present later on the path, is shadowed only because this directory was put first on purpose.
"""
from repconc_amd.faiss_compat import (METRIC_INNER_PRODUCT, METRIC_L2, IndexPQ, copy_array_to_vector, downcast_index,  # noqa: F401
                                      omp_set_num_threads, read_index, vector_to_array, write_index)
from repconc_amd.index import PQIndex as _PQIndex

IndexIVFPQ = _PQIndex          # the reference's 1-list IVFPQ wrapper is the PQ index itself here (from_pq_to_ivfpq)
__version__ = "0-repconc_amd-shim"
