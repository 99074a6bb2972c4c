#!/bin/bash
# developer helper: instruction-mix / fetch counters of bench.py for one workload (run on the GPU box)
# usage: tools/pmc_bench.sh <tag> <bench.py args...>     -> gpurun_out/pmc_<tag>/p{1,2,3}
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
B="python3 $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-also --no-tiers --no-traffic --no-steady $*"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES \
  --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAIT_ANY \
  --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE \
  --output-format csv -d $OUT/p3 -- $B > $OUT/p3.log 2>&1
echo "pmc done: $OUT"
