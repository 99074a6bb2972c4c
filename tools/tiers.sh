#!/bin/bash
# Run ON THE GPU BOX: the three timing tiers of SURVEY §8d for config-2-shaped
# input, plus bench lines for the other single-GPU configs.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tiers
mkdir -p $OUT
cd $ROOT
echo "nproc $(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)" 
echo "== (ii) H2D-inclusive (pinned double buffer)"
timeout -k 10 200 ./tools/kbench 2000000 150 0 0 0 0 1 1 | grep -E "h2d|mode=0"
QUACK_HIP_BATCH_MB=256 timeout -k 10 200 ./tools/kbench 4000000 150 0 0 0 0 1 1 | grep -E "h2d"
echo "== (iii) end-to-end CLI, 4M x 150 bp"
./tools/gen_fastq /tmp/e2e.fq.gz 4000000 150 150 2
./tools/gen_fastq /tmp/e2e.fq 4000000 150 150 2
ls -la /tmp/e2e.fq.gz /tmp/e2e.fq
for f in /tmp/e2e.fq.gz /tmp/e2e.fq; do
  for i in 1 2; do /usr/bin/time -f "$f wall %e s user %U s" ./quack_amd/host/quack -u $f > /tmp/e2e.svg; done
done
/usr/bin/time -f "gzip -dc wall %e s" gzip -dc /tmp/e2e.fq.gz > /dev/null
/usr/bin/time -f "oracle (CPU restatement) gz wall %e s" ./oracle/_build/quack_oracle time /tmp/e2e.fq.gz
/usr/bin/time -f "oracle (CPU restatement) plain wall %e s" ./oracle/_build/quack_oracle time /tmp/e2e.fq
echo "== bench cfg3 / cfg5"
timeout -k 10 400 python bench.py --workload cfg3 > $OUT/bench_cfg3.json 2>/dev/null; cat $OUT/bench_cfg3.json
timeout -k 10 400 python bench.py --workload cfg5 > $OUT/bench_cfg5.json 2>/dev/null; cat $OUT/bench_cfg5.json
