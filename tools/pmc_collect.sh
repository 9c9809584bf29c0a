#!/bin/bash
# Collect hardware counters per kernel for one command (GPU box).  One rocprofv3 --pmc pass per counter group
# (never combined with sys/runtime traces), outputs merged by tools/pmc_parse.py.
#   tools/pmc_collect.sh <out_json> -- <command...>
set -u
OUT=$1; shift; shift
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/pmc_work; rm -rf $W; mkdir -p $W
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS")
i=0
for g in "${GROUPS_[@]}"; do
  (cd /tmp && rocprofv3 --pmc $g --kernel-trace --output-format csv -d $W/g$i -o p -- "$@" > $W/g$i.log 2>&1) || echo "group $i failed (see $W/g$i.log)"
  i=$((i+1))
done
python $ROOT/tools/pmc_parse.py $W $OUT
