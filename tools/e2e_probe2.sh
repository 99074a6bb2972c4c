#!/bin/bash
# Run ON THE GPU BOX: e2e wall under environment variants:  tools/e2e_probe2.sh "VAR=.. VAR=.." ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
Q=./quack_amd/host/quack
[ -f /tmp/e2e.fq.gz ] || ./tools/gen_fastq /tmp/e2e.fq.gz 4000000 150 150 2
TIMEFORMAT="%R s wall, %U s user, %S s sys"
$Q -u /tmp/e2e.fq.gz > /tmp/e2e_ref.svg
for e in "$@"; do
  [ "$e" = "-" ] && ee="" || ee="$e"
  echo "== $e"
  for i in 1 2 3; do { time env $ee QUACK_VERBOSE=1 $Q -u /tmp/e2e.fq.gz > /tmp/e2e.svg; } 2>&1 | grep -v pgzip; cmp /tmp/e2e.svg /tmp/e2e_ref.svg; done
done
