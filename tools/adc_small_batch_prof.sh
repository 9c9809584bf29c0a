#!/bin/bash
# Per-kernel times of small-batch flat searches:  tools/adc_small_batch_prof.sh [nq=128] [k=200]   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
rm -rf /tmp/adcsb
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/adcsb -o ks -- python $ROOT/tools/adc_small_batch_prof.py ${1:-128} ${2:-200} > /tmp/adcsb.out 2>&1)
grep QPS /tmp/adcsb.out
python - "$(find /tmp/adcsb -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["Calls"]) >= 40 and "at::native" not in r["Name"]]
print(f"sum of kernels per search: {sum(float(r['TotalDurationNs']) for r in rows) / 41e3:.1f} us")
for r in rows[:14]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(5), f'{float(r["TotalDurationNs"])/41e3:9.1f} us/search')
PY
