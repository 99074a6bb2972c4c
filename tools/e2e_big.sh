#!/bin/bash
# Run ON THE GPU BOX: feed throughput on a file large enough that start-up does not matter (12M x 150 bp = 1.8 Gbases)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
Q=./quack_amd/host/quack
[ -f /tmp/big.fq.gz ] || ./tools/gen_fastq /tmp/big.fq.gz 12000000 150 150 7
ls -la /tmp/big.fq.gz | awk '{print $5, $9}'
TIMEFORMAT="%R s wall, %U s user, %S s sys"
$Q -u /tmp/big.fq.gz > /tmp/big_ref.svg
for e in "$@"; do
  [ "$e" = "-" ] && ee="" || ee="$e"
  echo "== $e"
  for i in 1 2; do { time env $ee QUACK_VERBOSE=1 $Q -u /tmp/big.fq.gz > /tmp/big.svg; } 2>&1 | grep -v "pgzip\|close +\|early:"; cmp /tmp/big.svg /tmp/big_ref.svg; done
done
