#!/bin/bash
# developer helper: PMC passes for one kbench configuration (run on the GPU box)
# usage: tools/pmc_probe.sh <tag> <kbench args...>
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
  --output-format csv -d $OUT/p1 -- $ROOT/tools/kbench "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_ATOMIC_RETURN \
  --output-format csv -d $OUT/p2 -- $ROOT/tools/kbench "$@" > $OUT/p2.log 2>&1
echo "pmc done: $OUT"
