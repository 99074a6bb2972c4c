#!/bin/bash
# developer helper: instruction-mix / fetch counters for one kbench binary + configuration (run on the GPU box)
# usage: tools/pmc_probe2.sh <tag> <kbench binary> <kbench args...>
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; BIN=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES \
  --output-format csv -d $OUT/p1 -- $ROOT/tools/$BIN "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQC_ICACHE_MISSES \
  --output-format csv -d $OUT/p2 -- $ROOT/tools/$BIN "$@" > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQC_ICACHE_REQ SQC_ICACHE_HITS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $OUT/p3 -- $ROOT/tools/$BIN "$@" > $OUT/p3.log 2>&1
echo "pmc done: $OUT"
