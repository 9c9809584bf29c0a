"""Static checks of bench.py against the driver's contract (no GPU): the legs added after the headline dictionary must
not reuse one of the contract's keys (a leg once named "warmup" replaced the integer the driver reads)."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def test_bench_legs_do_not_shadow_contract_keys():
    src = open(os.path.join(ROOT, "bench.py")).read()
    ast.parse(src)
    assigned = re.findall(r'out\["([A-Za-z0-9_]+)"\]\s*=', src)
    assert assigned, "no legs found - did the pattern change?"
    assert not (set(assigned) & CONTRACT), sorted(set(assigned) & CONTRACT)
    head = src[src.index("    out = {"):src.index("    if dist_check is not None:")]
    for key in CONTRACT:
        assert f'"{key}"' in head, key
    assert '"roofline": roofline' in head and 'out["cpu_baseline"]' in src


def test_bench_has_the_contract_flags():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
