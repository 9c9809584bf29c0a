"""A few rc_kmeans_stats calls at one size, for rocprofv3 / PMC passes:  python tools/kmeans_one.py [M] [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from repconc_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 48
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
g = torch.Generator(device="cuda:0").manual_seed(20228)
x = torch.randn((rows, 768), device="cuda:0", generator=g)
c = torch.randint(0, 256, (rows, M), dtype=torch.uint8, device="cuda:0", generator=g)
for _ in range(5):
    ops.kmeans_stats(x, c)
torch.cuda.synchronize()
