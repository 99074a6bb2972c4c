#!/bin/bash
# Run ON THE GPU BOX: one-thread inflate rates (round 3's decoder beside this one) and where the end-to-end run of
# config 2's file goes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out
L=gpurun_out/r4_e2e.log; : > $L
echo "nproc $(nproc); $(lscpu | grep -E 'Model name' | head -1)" >> $L
./tools/gen_fastq /tmp/s.fq.gz 4000000 150 150 2
echo "== one thread, 4M x 150 bp (0.6 Gbases, $(stat -c %s /tmp/s.fq.gz) bytes)" >> $L
[ -x tools/exp/inflate_bench_r3 ] && { echo "round 3 decoder:" >> $L; ./tools/exp/inflate_bench_r3 /tmp/s.fq.gz 2 >> $L; }
echo "this decoder:" >> $L; ./tools/inflate_bench /tmp/s.fq.gz 2 >> $L
echo "== config 2's file: 10M x 150 bp in 16 members" >> $L
for i in $(seq 0 15); do ./tools/gen_fastq /tmp/p$i.fq.gz 625000 150 150 $((2000+i)) & done; wait
cat /tmp/p*.fq.gz > /tmp/c2.fq.gz; rm /tmp/p*.fq.gz
TIMEFORMAT="%R s wall, %U s user, %S s sys"
for t in "" 8 16 24 32; do
  for i in 1 2; do echo -n "QUACK_THREADS=$t quack -u c2.fq.gz: " >> $L; { time QUACK_VERBOSE=1 QUACK_THREADS=$t ./quack_amd/host/quack -u /tmp/c2.fq.gz > /tmp/c2.svg 2> /tmp/c2.err; } 2>> $L; grep -E "accumulators|close" /tmp/c2.err >> $L; done
done
echo -n "QUACK_ZLIB=1: " >> $L; { time QUACK_ZLIB=1 ./quack_amd/host/quack -u /tmp/c2.fq.gz > /tmp/c2z.svg; } 2>> $L; cmp /tmp/c2.svg /tmp/c2z.svg >> $L
cat $L
