cd $GRAFT_REPO_ROOT; O=gpurun_out/r02h; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu --timeout 900 -k "ivf" 2>&1 | tail -30 > $O/pytest_ivf.txt; grep -E "^FAILED|^ERROR|passed|failed|Error|assert" $O/pytest_ivf.txt | tail -12
python bench.py --no-cpu > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d.get("ivf"))
PY
