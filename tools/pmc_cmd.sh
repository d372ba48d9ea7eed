#!/bin/bash
# PMC passes (counters only, one process per pass) for an arbitrary python tool of this repo.
# usage: tools/pmc_cmd.sh <tag> <script.py> [args...]   -> gpurun_out/pmc_<tag>/summary.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
SCRIPT=$1; shift
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for counters in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  ${PMC_TRAFFIC:+"FETCH_SIZE"} ${PMC_TRAFFIC:+"WRITE_SIZE"} ; do
  i=$((i+1))
  rocprofv3 --pmc $counters --output-format csv -d $OUT/pass$i -- python3 $ROOT/$SCRIPT "$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; }
done
python3 $ROOT/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
