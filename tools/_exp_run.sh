cd $GRAFT_REPO_ROOT; O=gpurun_out/r02l; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu --timeout 600 -k "graph or golden_codes or native_rccl" 2>&1 | tail -3
python bench.py --no-cpu > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches_timed"])
print(d["per_rank_6144"]["value"], d["per_rank_6144"]["ms_per_step"], d["per_rank_6144"]["roofline"]["frac"], d["per_rank_6144"]["roofline"]["avg_launch_ms"])
print(d["adc"]["value"], d["ivf"]["queries_per_sec"])
PY
