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
  /* what a pinflate.c worker does with a slice, stage by stage, on one thread: the 16-bit (marker) decode, markers -> bytes,
   * CRC-32, line index — in blocks of 16 MiB of output like a 4 MiB slice's */
  {
    enum { OUT16 = 16 << 20 };
    uint16_t *b16 = malloc((HIST + OUT16 + 64) * sizeof *b16);
    uint8_t *b8 = malloc(OUT16 + 64), *lut = malloc(65536);
    uint32_t *nl = NULL;
    size_t cap_nl = 0;
    for (unsigned i = 0; i < 65536; i++) lut[i] = (uint8_t)i;
    qkh_inflate *z = malloc(sizeof *z);
    qkh_inflate_init(z, data, st.st_size);
    size_t total = 0, hist = 0;
    double t_dec = 0, t_res = 0, t_crc = 0, t_nl = 0;
    uint32_t crc = 0;
    for (;;) {
      double t0 = now();
      size_t n = 0;
      for (;;) {
        z->tl_n = 0;
        long got = qkh_inflate_read16(z, b16 + HIST + n, OUT16 - n, hist + n);
        if (got > 0) n += (size_t)got;
        if (n == OUT16 || (got <= 0 && !qkh_inflate_log_full(z))) break;
      }
      double t1 = now();
      if (!n) break;
      extern void qkh_resolve16(const uint16_t *src, uint8_t *dst, size_t n, const uint8_t *lut);
      qkh_resolve16(b16 + HIST, b8, n, lut);
      double t2 = now();
      crc = qkh_crc32(crc, b8, n);
      double t3 = now();
      size_t lines = qkh_index_lines(b8, n, &nl, &cap_nl);
      double t4 = now();
      (void)lines;
      t_dec += t1 - t0, t_res += t2 - t1, t_crc += t3 - t2, t_nl += t4 - t3;
      total += n;
      size_t keep = n < HIST ? n : HIST;
      memmove(b16 + HIST - keep, b16 + HIST + n - keep, keep * sizeof *b16);
      hist = hist + n < HIST ? hist + n : HIST;
    }
    printf("worker stages: %zu bytes; decode16 %.3f s = %.1f MB/s, resolve %.3f s = %.1f MB/s, crc32 %.3f s = %.1f MB/s, line index %.3f s = %.1f MB/s; all %.1f MB/s (crc %08x)\n",
           total, t_dec, total / t_dec / 1e6, t_res, total / t_res / 1e6, t_crc, total / t_crc / 1e6, t_nl, total / t_nl / 1e6,
           total / (t_dec + t_res + t_crc + t_nl) / 1e6, crc);
  }
  /* what finding a block start costs a worker: from every 4 MiB of the compressed file on, like pinflate.c's slices */
  {
    qkh_inflate *z = malloc(sizeof *z);
    double t0 = now();
    unsigned found = 0, tried = 0;
    uint64_t skipped = 0;
    for (size_t at = (size_t)4 << 20; at + ((size_t)1 << 20) < (size_t)st.st_size; at += (size_t)4 << 20) {
      const int64_t bit = qkh_inflate_find_block(data, st.st_size, (uint64_t)at * 8u, (uint64_t)st.st_size * 8u, z);
      tried++;
      if (bit >= 0) {
        found++;
        skipped += (uint64_t)bit - (uint64_t)at * 8u;
      }
    }
    double dt = now() - t0;
    if (tried)
      printf("find_block   : %u of %u slice starts found in %.4f s = %.3f ms per slice, %.1f KiB of compressed data skipped on average\n", found, tried, dt,
             dt / tried * 1e3, found ? skipped / 8.0 / found / 1024.0 : 0.0);
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
