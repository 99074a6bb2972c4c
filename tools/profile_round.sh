#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats of the bench command and
# separate PMC passes for HBM traffic.  Output under gpurun_out/prof_<tag>/;
# tools/summarize_profile.py turns it into the files committed in profiles/.
#   tools/profile_round.sh <tag> [bench args...]
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-tiers --no-traffic --no-steady $*"
echo "== kernel trace"
# same step counts as the default bench line, so that the two averages are comparable
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --no-tiers --no-traffic --no-steady $* > $OUT/trace.log 2>&1
echo "== pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "== pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
echo "== pmc raw TCC"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH > $OUT/pmc_tcc.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc2 -- $BENCH > $OUT/pmc_tcc2.log 2>&1
echo "== pmc SQ (instruction mix, busy / wait cycles, LDS)"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES \
  --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAIT_ANY \
  --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE \
  --output-format csv -d $OUT/sq3 -- $BENCH > $OUT/sq3.log 2>&1
if [ -x $ROOT/tools/kbench ] && [ -z "$*" ]; then
  echo "== calibration: loads-only kernel (every byte loaded exactly once), same access pattern"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_calib -- $ROOT/tools/kbench 10000000 150 0 0 0 1 1 > $OUT/pmc_calib.log 2>&1
fi
echo "profile done: $OUT"
