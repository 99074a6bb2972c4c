#!/bin/bash
# Run ON THE GPU BOX (host only): where the feed's time goes on config 2's file without any GPU call
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out; L=gpurun_out/feed_bench.log; : > $L
cc -O3 -o tools/feed_bench tools/feed_bench.c -Iinclude -Iquack_amd/host -Lquack_amd -lquack_host -lquack_hip -Wl,-rpath,'$ORIGIN/../quack_amd' || exit 1
for i in $(seq 0 15); do ./tools/gen_fastq /tmp/p$i.fq.gz 625000 150 150 $((2000+i)) & done; wait
cat /tmp/p*.fq.gz > /tmp/c2.fq.gz; rm /tmp/p*.fq.gz
echo "nproc $(nproc)" >> $L
for dt in 16 32 64; do for tt in 1 4 8; do
  for i in 1 2; do echo -n "decoders $dt tokenizer $tt: " >> $L; QUACK_THREADS=$dt QUACK_TOKENIZER_THREADS=$tt ./tools/feed_bench /tmp/c2.fq.gz >> $L; done
done; done
gzip -dc /tmp/c2.fq.gz > /tmp/c2.fq
for tt in 1 4 8; do for i in 1 2; do echo -n "plain file, tokenizer $tt: " >> $L; QUACK_TOKENIZER_THREADS=$tt ./tools/feed_bench /tmp/c2.fq >> $L; done; done
cat $L
