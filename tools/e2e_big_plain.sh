#!/bin/bash
# Run ON THE GPU BOX: the tokenizer without inflate in front of it (12M x 150 bp uncompressed, from the page cache)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
Q=./quack_amd/host/quack
[ -f /tmp/big.fq ] || ./tools/gen_fastq /tmp/big.fq 12000000 150 150 7
ls -la /tmp/big.fq | awk '{print $5, $9}'
cat /tmp/big.fq > /dev/null
TIMEFORMAT="%R s wall, %U s user, %S s sys"
$Q -u /tmp/big.fq > /tmp/big_ref.svg
for e in "$@"; do
  [ "$e" = "-" ] && ee="" || ee="$e"
  echo "== $e"
  for i in 1 2; do { time env $ee QUACK_VERBOSE=1 QUACK_FULL_TEARDOWN=1 $Q -u /tmp/big.fq > /tmp/big.svg; } 2>&1 | grep -v "pgzip\|close +\|early:"; cmp /tmp/big.svg /tmp/big_ref.svg; done
done
