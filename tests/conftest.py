import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden_cases():
    import glob
    return sorted(os.path.basename(p)[len("quantize_"):-4] for p in glob.glob(os.path.join(GOLDEN, "quantize_*.npz")))


def load_case(name):
    import numpy as np
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, f"quantize_{name}.npz"))
    M, B, kind = int(g["M"]), int(g["B"]), str(g["kind"])
    seed, x, C = synth.golden_case_inputs(name, M, B, kind)
    assert seed == int(g["seed"])
    return g, x, C


def load_headline(kind, M=48):
    """tests/golden/headline_b49152_m<M>_<kind>.npz (oracle/gen_golden.py --headline [--headline-m M]): the REFERENCE's codes
    of one whole 49 152 x 768 training batch at M = 48 (BASELINE configs[1]) or M = 24 (c5), run as four column slices of
    192 columns.  Returns (x, C, constrained codes, nearest codes)."""
    import zlib
    import numpy as np
    from oracle import synth
    g = np.load(os.path.join(GOLDEN, f"headline_b49152_m{M}_{kind}.npz"))
    B, M = int(g["B"]), int(g["M"])
    x = synth.clustered_embeddings(int(g["x_seed"]), B, n_clusters=int(g["n_clusters"]))
    C = g["centroids"] if "centroids" in g.files else synth.sample_centroids(int(g["c_seed"]), x, M)
    assert zlib.crc32(np.ascontiguousarray(C).tobytes()) == int(g["centroids_crc"])
    con = g["codes_constrained"]
    return x, C, con, con ^ g["codes_nearest_xor_constrained"]
