/* inflate_bench — one-thread decode rate of quack_amd/host/inflate_fast.c against zlib on the same .gz
 *   cc -O3 -o tools/inflate_bench tools/inflate_bench.c quack_amd/host/inflate_fast.c quack_amd/host/crc32_fold.c -Iquack_amd/host -lz
 *   tools/inflate_bench file.fq.gz [passes] */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include "inflate_fast.h"

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv) {
  int passes = argc > 2 ? atoi(argv[2]) : 3;
  int fd = open(argv[1], O_RDONLY);
  struct stat st;
  fstat(fd, &st);
  const uint8_t *data = mmap(NULL, st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
  enum { BLOCK = 4 << 20, HIST = 32768 };
  uint8_t *buf = malloc(HIST + BLOCK + 64);
  for (int p = 0; p < passes; p++) {
    qkh_inflate *z = malloc(sizeof *z);
    qkh_inflate_init(z, data, st.st_size);
    size_t total = 0, hist = 0;
    uint32_t crc = 0;
    double t0 = now();
    for (;;) {
      long n = qkh_inflate_read(z, buf + HIST, BLOCK, hist);
      if (n <= 0 && !qkh_inflate_log_full(z)) break;
      if (n > 0) {
        total += n;
        crc ^= buf[HIST + (n >> 1)];
        size_t keep = (size_t)n < HIST ? (size_t)n : HIST;
        memmove(buf + HIST - keep, buf + HIST + n - keep, keep);   /* (n < HIST only at the very end) */
        hist = hist + n < HIST ? hist + n : HIST;
      }
    }
    double dt = now() - t0;
    printf("inflate_fast: %zu bytes in %.3f s = %.1f MB/s (x%02x)\n", total, dt, total / dt / 1e6, crc);
    free(z);
  }
  {
    gzFile g = gzopen(argv[1], "rb");
    gzbuffer(g, 1 << 20);
    size_t total = 0;
    double t0 = now();
    int n;
    while ((n = gzread(g, buf, BLOCK)) > 0) total += n;
    double dt = now() - t0;
    gzclose(g);
    printf("zlib gzread : %zu bytes in %.3f s = %.1f MB/s\n", total, dt, total / dt / 1e6);
  }
  return 0;
}
