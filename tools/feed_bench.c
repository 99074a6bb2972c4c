/* feed_bench — the host feed without the GPU: source (inflate threads) -> tokenizer -> two scratch batch buffers, timed.
 *   cc -O3 -o tools/feed_bench tools/feed_bench.c -Iinclude -Iquack_amd/host -Lquack_amd -lquack_host -lquack_hip -Wl,-rpath,'$ORIGIN/../quack_amd'
 *   QUACK_THREADS=32 QUACK_TOKENIZER_THREADS=4 tools/feed_bench file.fq.gz [slot MiB] */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "quack_host.h"

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv) {
  const size_t cap = (size_t)(argc > 2 ? atoi(argv[2]) : 32) << 20, cap_reads = cap / 32 + 1024;
  uint8_t *seq[2], *qual[2];
  uint64_t *off = malloc((cap_reads + 1) * sizeof *off);
  for (int i = 0; i < 2; i++) {
    seq[i] = malloc(cap + 64);
    qual[i] = malloc(cap + 64);
    memset(seq[i], 1, cap);
    memset(qual[i], 1, cap);
  }
  const double t0 = now();
  qkh_reader *r = qkh_reader_open(argv[1]);
  if (!r) return 1;
  uint64_t reads = 0, bases = 0;
  int turn = 0;
  while (!qkh_reader_done(r)) {
    uint64_t total = 0;
    uint32_t uni = 0;
    int64_t n = qkh_reader_fill(r, seq[turn], qual[turn], off, cap, cap_reads, &total, &uni);
    if (n < 0) return 2;
    reads += (uint64_t)n;
    bases += total;
    turn ^= 1;
  }
  const double dt = now() - t0;
  printf("%llu reads, %llu bases in %.3f s = %.2f Gbases/s (%.1f GB/s of FASTQ text at ~2.2 bytes per base)\n", (unsigned long long)reads,
         (unsigned long long)bases, dt, bases / dt / 1e9, bases * 2.2 / dt / 1e9);
  qkh_reader_close(r);
  return 0;
}
