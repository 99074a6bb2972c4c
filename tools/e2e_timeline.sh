#!/bin/bash
# Run ON THE GPU BOX: where an end-to-end run of config 2's file (10M x 150 bp, 16 gzip members) and of a 4M-read file goes —
# QUACK_VERBOSE's timeline of the pipeline — for this build and, when tools/exp/old_host/libquack_host.so exists, the build before
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out
L=gpurun_out/e2e_timeline.log; : > $L
echo "nproc $(nproc); $(lscpu | grep -E 'Model name' | head -1)" >> $L
for i in $(seq 0 15); do ./tools/gen_fastq /tmp/p$i.fq.gz 625000 150 150 $((2000+i)) & done; wait
cat /tmp/p*.fq.gz > /tmp/c2.fq.gz
cat /tmp/p0.fq.gz /tmp/p1.fq.gz /tmp/p2.fq.gz /tmp/p3.fq.gz /tmp/p4.fq.gz /tmp/p5.fq.gz > /tmp/c2s.fq.gz; rm /tmp/p*.fq.gz
TIMEFORMAT="%R s wall"
one() {   # label, file, extra args
  echo -n "$1 $2 $3: " >> $L; { time QUACK_VERBOSE=1 ./quack_amd/host/quack -u $2 $3 > /tmp/o.svg 2> /tmp/o.err; } 2>> $L; grep -E "accumulators" /tmp/o.err | sed 's/.*fq.gz: /    /' >> $L
}
for rep in 1 2 3 4; do
  [ -f tools/exp/old_host/libquack_host.so ] && LD_LIBRARY_PATH=tools/exp/old_host:quack_amd one "before" /tmp/c2.fq.gz
  one "now   " /tmp/c2.fq.gz
done
cp /tmp/o.svg /tmp/now.svg
LD_LIBRARY_PATH=tools/exp/old_host:quack_amd ./quack_amd/host/quack -u /tmp/c2.fq.gz > /tmp/old.svg 2>/dev/null; cmp /tmp/now.svg /tmp/old.svg && echo "same SVG" >> $L
for rep in 1 2 3; do
  [ -f tools/exp/old_host/libquack_host.so ] && LD_LIBRARY_PATH=tools/exp/old_host:quack_amd one "before" /tmp/c2s.fq.gz
  one "now   " /tmp/c2s.fq.gz
done
for rep in 1 2; do
  [ -f tools/exp/old_host/libquack_host.so ] && LD_LIBRARY_PATH=tools/exp/old_host:quack_amd one "before" /tmp/c2.fq.gz "-a tests/golden/inputs/adapters.fa"
  one "now   " /tmp/c2.fq.gz "-a tests/golden/inputs/adapters.fa"
done
cat $L
