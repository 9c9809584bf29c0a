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


def test_bench_measures_what_survey_8d_asks_for():
    """SURVEY 8d / VERDICT r4 item 4: a second ADC leg on codes from the index build (with the screen's survivors per query),
    the `import faiss` probe with the port as fall-back, CPU baselines of kind port / torch / faiss on a first-touched copy."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for needle in ('["index_built_codes"]', "screen_survivors_per_query", "import faiss", '"kind": "torch"', '"kind": "faiss"',
                   "first_touch_copy", "adc_search(sl, cent, qc, k, tile=0)", "rc_solve_num_chains_on"):
        assert needle in src, needle


def test_bench_has_the_contract_flags():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src


def test_bench_spawns_its_own_ranks_when_typed_without_a_launcher():
    """`python bench.py --gpus N` with WORLD_SIZE unset (how the driver types it) must not die with 'launch with
    torch.distributed.run': it re-executes itself once per rank; with too few GPUs it prints ONE {"skipped": ...} JSON
    line and exits 0 (checked here on the GPU-less box)."""
    import json
    import subprocess
    import sys
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "def spawn_ranks(" in src and '"WORLD_SIZE" not in os.environ' in src
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RC_BENCH_SHARE_GPU")}
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert "skipped" in line and line["n_gpus"] == 2 and line["value"] is None
