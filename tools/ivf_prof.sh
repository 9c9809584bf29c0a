#!/bin/bash
# Per-kernel times of the IVF search (tools/ivf_bench.py M):  tools/ivf_prof.sh [M]   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
rm -rf /tmp/ivfks
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ivfks -o ks -- python $ROOT/tools/ivf_bench.py ${1:-96} > /tmp/ivf_prof.out 2>&1)
grep "QPS" /tmp/ivf_prof.out
python - "$(find /tmp/ivfks -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:30]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(5), f'{float(r["AverageNs"])/1e3:10.1f} us', r["Percentage"])
PY
