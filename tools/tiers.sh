#!/bin/bash
# Run ON THE GPU BOX: the timing tiers of SURVEY §8d for config-2-shaped input.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
echo "nproc $(nproc)"; lscpu | grep -E "Model name" 
echo "== (ii) H2D-inclusive (pinned double buffer)"
timeout -k 10 200 ./tools/kbench 2000000 150 0 0 12 0 1 1 | grep -E "h2d"
echo "== (iii) end-to-end CLI, 4M x 150 bp (0.6 Gbases)"
./tools/gen_fastq /tmp/e2e.fq.gz 4000000 150 150 2
./tools/gen_fastq /tmp/e2e.fq 4000000 150 150 2
./tools/gen_fastq /tmp/e2e_R2.fq.gz 4000000 150 150 5 2 30
./tools/gen_fastq /tmp/e2e.fq.bgz 4000000 150 150 2
ls -la /tmp/e2e.fq.gz /tmp/e2e.fq | awk '{print $5, $9}'
TIMEFORMAT="%R s wall, %U s user"
for f in /tmp/e2e.fq.gz /tmp/e2e.fq.bgz /tmp/e2e.fq; do
  for i in 1 2; do echo -n "quack -u $f : "; { time ./quack_amd/host/quack -u $f > /tmp/e2e.svg; } 2>&1; done
done
for t in 1 2 4 8 16 32; do echo -n "QUACK_THREADS=$t quack -u gz : "; { time QUACK_VERBOSE=1 QUACK_THREADS=$t ./quack_amd/host/quack -u /tmp/e2e.fq.gz > /tmp/e2e_t.svg; } 2>&1; cmp /tmp/e2e.svg /tmp/e2e_t.svg || echo "SVG DIFFERS"; done
echo -n "QUACK_NO_PGZIP=1 quack -u gz : "; { time QUACK_NO_PGZIP=1 ./quack_amd/host/quack -u /tmp/e2e.fq.gz > /tmp/e2e_t.svg; } 2>&1; cmp /tmp/e2e.svg /tmp/e2e_t.svg || echo "SVG DIFFERS"
echo -n "quack -1 gz -2 gz (paired, 1.2 Gbases): "; { time ./quack_amd/host/quack -1 /tmp/e2e.fq.gz -2 /tmp/e2e_R2.fq.gz > /tmp/e2e.svg; } 2>&1
echo -n "gzip -dc: "; { time gzip -dc /tmp/e2e.fq.gz > /dev/null; } 2>&1
echo -n "oracle (CPU restatement) gz: "; { time ./oracle/_build/quack_oracle time /tmp/e2e.fq.gz; } 2>&1
echo -n "oracle (CPU restatement) plain: "; { time ./oracle/_build/quack_oracle time /tmp/e2e.fq; } 2>&1
