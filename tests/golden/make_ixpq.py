#!/usr/bin/env python3
"""Assembles tests/golden/ixpq_d8_m2_n5.faissindex byte by byte in the order Faiss 1.7.x `write_index` emits an IndexPQ
(impl/index_write.cpp: fourcc, write_index_header, write_ProductQuantizer, WRITEVECTOR(codes), search_type, encode_signs,
polysemous_ht) — written independently of repconc_amd/faiss_io.py so that the reader is checked against a second
statement of the layout.  No Faiss build is reachable offline: the layout is restated from the Faiss sources as
remembered, NOT verified against a Faiss-written file.  d = 8, M = 2, nbits = 8, 5 rows, METRIC_INNER_PRODUCT.
    python tests/golden/make_ixpq.py"""
import os
import struct

import numpy as np

d, M, nbits, n = 8, 2, 8, 5
rng = np.random.default_rng(424242)
centroids = rng.standard_normal((M, 1 << nbits, d // M)).astype("<f4")       # ProductQuantizer::centroids [m][k][j]
codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
b = bytearray()
b += b"IxPq"                                      # fourcc("IxPq"), uint32 little endian = these four bytes
b += struct.pack("<i", d)                         # Index::d                int
b += struct.pack("<q", n)                         # Index::ntotal           idx_t (int64)
b += struct.pack("<q", 1 << 20) * 2               # two dummies             idx_t
b += struct.pack("<?", True)                      # Index::is_trained       bool (1 byte)
b += struct.pack("<i", 0)                         # Index::metric_type      METRIC_INNER_PRODUCT = 0
b += struct.pack("<Q", d) + struct.pack("<Q", M) + struct.pack("<Q", nbits)   # ProductQuantizer d, M, nbits: size_t
b += struct.pack("<Q", centroids.size) + centroids.tobytes()                  # WRITEVECTOR(pq.centroids)
b += struct.pack("<Q", codes.size) + codes.tobytes()                          # WRITEVECTOR(codes)
b += struct.pack("<i", 0)                         # search_type             ST_PQ
b += struct.pack("<?", False)                     # encode_signs            bool
b += struct.pack("<i", 0)                         # polysemous_ht           int
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ixpq_d8_m2_n5.faissindex")
open(out, "wb").write(bytes(b))
np.savez(os.path.join(os.path.dirname(out), "ixpq_d8_m2_n5_expected.npz"), centroids=centroids, codes=codes)
print(out, len(b), "bytes")
