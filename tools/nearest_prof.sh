#!/bin/bash
# Per-kernel times of the index build (tools/nearest_bench.py N M):  tools/nearest_prof.sh [N] [M]   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
rm -rf /tmp/nnks
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nnks -o ks -- python $ROOT/tools/nearest_bench.py ${1:-1048576} ${2:-48} > /tmp/nn_prof.out 2>&1)
grep "mfma:" /tmp/nn_prof.out
python - "$(find /tmp/nnks -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if "assign" in r["Name"]:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(4), f'{float(r["AverageNs"])/1e3:10.1f} us avg', f'{float(r["MaxNs"])/1e3:10.1f} us max')
PY
