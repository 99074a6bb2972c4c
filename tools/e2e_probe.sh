#!/bin/bash
# Run ON THE GPU BOX: where the wall time of one CLI run goes (process load, HIP start-up, feed, teardown).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
Q=./quack_amd/host/quack
[ -f /tmp/e2e.fq.gz ] || ./tools/gen_fastq /tmp/e2e.fq.gz 4000000 150 150 2
TIMEFORMAT="%R s wall, %U s user, %S s sys"
echo -n "usage only (process load): "; { time $Q > /dev/null; } 2>&1
echo -n "usage only (process load): "; { time $Q > /dev/null; } 2>&1
LD_DEBUG=statistics $Q 2>&1 >/dev/null | grep -E "total startup|relocation|load" | head -5
for i in 1 2 3; do { time QUACK_VERBOSE=1 $Q -u /tmp/e2e.fq.gz "$@" > /tmp/e2e.svg; } 2>&1; done
