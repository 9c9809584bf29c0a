cd $GRAFT_REPO_ROOT; O=gpurun_out/r02k; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -k "ivf" 2>&1 | tail -40 > $O/pytest_ivf.txt; grep -E "^FAILED|^ERROR|passed|failed|Error|assert " $O/pytest_ivf.txt | tail -12
python tools/ivf_bench.py 96 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $GRAFT_REPO_ROOT/tools/ivf_bench.py 96 > /dev/null 2>&1; python - <<PY
import csv,glob
f=glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), f'{float(r["AverageNs"])/1e3:10.1f} us', r["Percentage"])
PY
