#!/bin/bash
# Run ON THE GPU BOX (host only): where the feed's time goes on config 2's file without any GPU call
#   tools/feed_bench.sh            decoder-thread sweep on the .gz and the plain file
#   tools/feed_bench.sh slices     slice size of the multi-threaded inflate (QUACK_PGZIP_CHUNK_KB) x decoder threads
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out; L=gpurun_out/feed_bench.log; : > $L
make tools/feed_bench > /dev/null || exit 1
for i in $(seq 0 15); do ./tools/gen_fastq /tmp/p$i.fq.gz 625000 150 150 $((2000+i)) & done; wait
cat /tmp/p*.fq.gz > /tmp/c2.fq.gz; rm /tmp/p*.fq.gz
echo "nproc $(nproc)" >> $L
if [ "$1" = slices ]; then
  for kb in 512 1024 2048 4096 8192; do for dt in 16 32; do
    for i in 1 2; do echo -n "slice $kb KiB, decoders $dt: " >> $L; QUACK_VERBOSE=1 QUACK_PGZIP_CHUNK_KB=$kb QUACK_THREADS=$dt ./tools/feed_bench /tmp/c2.fq.gz 2>&1 | grep -E "bases|slices" | tr '\n' ' ' >> $L; echo >> $L; done
  done; done
else
  for dt in 16 32 64; do
    for i in 1 2; do echo -n "decoders $dt: " >> $L; QUACK_THREADS=$dt ./tools/feed_bench /tmp/c2.fq.gz >> $L; done
  done
  gzip -dc /tmp/c2.fq.gz > /tmp/c2.fq
  for i in 1 2; do echo -n "plain file: " >> $L; ./tools/feed_bench /tmp/c2.fq >> $L; done
fi
cat $L
