#!/bin/bash
# Per-kernel times of the flat ADC search alone:  tools/adc_prof.sh [M ...]   (GPU box; prints the top kernels)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
rm -rf /tmp/adcks
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/adcks -o ks -- python $ROOT/tools/adc_quick_bench.py ${@:-48} > /tmp/adc_prof.out 2>&1)
grep QPS /tmp/adc_prof.out
python - "$(find /tmp/adcks -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:18]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), f'{float(r["AverageNs"])/1e3:10.1f} us', r["Percentage"])
PY
