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
