#!/bin/bash
# Per-kernel times of ONE IVF configuration (tools/ivf_one.py nprobe):  tools/ivf_one_prof.sh <nprobe> [out.txt] [nq = 1200]   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
rm -rf /tmp/ivf1ks
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ivf1ks -o ks -- python $ROOT/tools/ivf_one.py ${1:-128} ${3:-1200} > /tmp/ivf1_prof.out 2>&1)
python - "$(find /tmp/ivf1ks -name '*kernel_stats.csv' | head -1)" <<'PY' | tee ${2:-/dev/null}
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["Calls"]) >= 20 and "at::native" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 20e3
print(f"sum of kernels per search: {tot:.1f} us")
for r in rows[:32]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), f'{float(r["TotalDurationNs"])/20e3:10.1f} us/search')
PY
