"""Child of bench.py's live HBM-traffic measurement (run under `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`): two
constrained assignments of one 49152 x 768 batch (M = 48, eps 0.003, T = 100) with eager sweep launches (RC_GRAPH=0), so that
every launch of sk_sweep2_kernel<2, true> is a dispatch of its own in the counter CSV."""
import os
import sys

os.environ["RC_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from repconc_amd import ops  # noqa: E402

B, D, M, K = int(sys.argv[1]) if len(sys.argv) > 1 else 49152, 768, 48, 256
g = torch.Generator(device="cuda:0").manual_seed(20220)
x = torch.randn((B, D), device="cuda:0", generator=g)
C = x[torch.randperm(B, device="cuda:0", generator=g)[:K]].reshape(K, M, D // M).transpose(0, 1).contiguous()
for _ in range(2):
    codes, flags = ops.assign_sinkhorn(x, C, 0.003, 100, torch.uint8)
torch.cuda.synchronize()
assert int(flags.item()) == 0
