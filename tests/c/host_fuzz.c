/*
 * host_fuzz.c — AddressSanitizer/UBSan fuzz of the host C code (CPU only).
 *
 * Links the PRODUCT tokenizer + gzip decoder (quack_amd/host/reader.c,
 * inflate_fast.c) and, as the checker, the ORACLE tokenizer
 * (oracle/quack_oracle.c).  Generates hostile FASTQ-ish text, writes it plain
 * and gzip-compressed (several members, assorted levels), tokenises each file
 * with both, in batches of awkward sizes, and compares every record.
 *
 *   host_fuzz [iterations] [seed]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include "quack_host.h"
#include "quack_oracle.h"

/* reader.c pulls in the C-ABI header only; no accumulator call is made here,
 * but pipeline.c is not linked, so nothing else is needed */

static uint64_t rs;
static uint64_t rnd(void) {
  uint64_t z = (rs += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static size_t make_text(char *buf, size_t cap) {
  static const char *alpha = "ACGTNacgtn@+>I5#! \r\t";
  size_t n = 0;
  int mode = (int)(rnd() % 4);
  int recs = (int)(rnd() % 60);
  for (int r = 0; r < recs && n + 1200 < cap; r++) {
    if (mode == 0 || rnd() % 7 == 0) { /* noise lines */
      int lines = (int)(rnd() % 4);
      for (int l = 0; l < lines; l++) {
        int len = (int)(rnd() % 14);
        for (int i = 0; i < len; i++) buf[n++] = alpha[rnd() % 20];
        if (rnd() % 5 == 0) buf[n++] = '\r';
        buf[n++] = '\n';
      }
      continue;
    }
    int len = (int)(rnd() % (mode == 3 ? 900 : 70));
    const char *nl = (rnd() % 4 == 0) ? "\r\n" : "\n";
    n += (size_t)sprintf(buf + n, "@r%d%s%s", r, rnd() % 2 ? " comment" : "", nl);
    int split = (len > 4 && rnd() % 3 == 0) ? (int)(rnd() % len) : -1;
    for (int i = 0; i < len; i++) {
      if (i == split) n += (size_t)sprintf(buf + n, "%s%s", nl, rnd() % 4 == 0 ? nl : "");
      buf[n++] = "ACGTN"[rnd() % 5];
    }
    n += (size_t)sprintf(buf + n, "%s+%s", nl, nl);
    int qlen = (rnd() % 11 == 0) ? (int)(rnd() % (len + 3)) : len; /* sometimes a wrong length */
    int qsplit = (qlen > 4 && rnd() % 3 == 0) ? (int)(rnd() % qlen) : -1;
    for (int i = 0; i < qlen; i++) {
      if (i == qsplit) n += (size_t)sprintf(buf + n, "%s", nl);
      buf[n++] = (char)(33 + rnd() % 60);
    }
    if (!(r == recs - 1 && rnd() % 3 == 0)) n += (size_t)sprintf(buf + n, "%s", nl);
  }
  return n;
}

static void bgzf_block(FILE *f, const unsigned char *data, size_t n, int level) {
  unsigned char out[70000];
  z_stream s;
  memset(&s, 0, sizeof s);
  deflateInit2(&s, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
  s.next_in = (unsigned char *)data;
  s.avail_in = (uInt)n;
  s.next_out = out;
  s.avail_out = sizeof out;
  deflate(&s, Z_FINISH);
  size_t clen = sizeof out - s.avail_out;
  deflateEnd(&s);
  unsigned bsize = (unsigned)(18 + clen + 8 - 1);
  unsigned long crc = crc32(crc32(0, NULL, 0), data, (uInt)n);
  unsigned char h[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 255), (unsigned char)(bsize >> 8)};
  unsigned char t[8] = {(unsigned char)crc, (unsigned char)(crc >> 8), (unsigned char)(crc >> 16), (unsigned char)(crc >> 24),
                        (unsigned char)n, (unsigned char)(n >> 8), (unsigned char)(n >> 16), (unsigned char)(n >> 24)};
  fwrite(h, 1, 18, f);
  fwrite(out, 1, clen, f);
  fwrite(t, 1, 8, f);
}

static void write_file(const char *path, const char *data, size_t n, int gz) {
  FILE *f = fopen(path, "wb");
  if (!gz) {
    fwrite(data, 1, n, f);
    fclose(f);
    return;
  }
  if (gz == 2) { /* BGZF: members of 1..6000 bytes, sometimes an ordinary gzip tail */
    size_t done = 0, tail = rnd() % 4 == 0 ? (size_t)(rnd() % (n + 1)) : n;
    while (done < tail) {
      size_t k = 1 + (size_t)(rnd() % 6000);
      if (k > tail - done) k = tail - done;
      bgzf_block(f, (const unsigned char *)data + done, k, (int)(rnd() % 10));
      done += k;
    }
    if (rnd() % 2) bgzf_block(f, (const unsigned char *)data, 0, 6);
    fclose(f);
    if (done < n) { /* ordinary member appended after the BGZF part */
      gzFile g = gzopen(path, "ab6");
      gzwrite(g, data + done, (unsigned)(n - done));
      gzclose(g);
    }
    return;
  }
  fclose(f);
  /* 1-3 gzip members at assorted levels / strategies */
  int members = 1 + (int)(rnd() % 3);
  size_t done = 0;
  f = fopen(path, "wb");
  for (int m = 0; m < members; m++) {
    size_t part = m == members - 1 ? n - done : (size_t)(rnd() % (n - done + 1));
    z_stream s;
    memset(&s, 0, sizeof s);
    static const int strategies[4] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE};
    deflateInit2(&s, (int)(rnd() % 10), Z_DEFLATED, 31, 1 + (int)(rnd() % 9), strategies[rnd() % 4]);
    size_t cap = deflateBound(&s, part) + 64;
    unsigned char *out = malloc(cap);
    s.next_in = (unsigned char *)data + done;
    s.avail_in = (uInt)part;
    s.next_out = out;
    s.avail_out = (uInt)cap;
    deflate(&s, Z_FINISH);
    fwrite(out, 1, cap - s.avail_out, f);
    deflateEnd(&s);
    free(out);
    done += part;
  }
  fclose(f);
}

static int compare(const char *path, size_t cap_bytes, size_t cap_reads) {
  oracle_reader *o = oracle_reader_open(path);
  qkh_reader *p = qkh_reader_open(path);
  uint8_t *seq = malloc(cap_bytes + 16), *qual = malloc(cap_bytes + 16);
  uint64_t *off = malloc((cap_reads + 1) * sizeof *off);
  int bad = 0;
  if (!o || !p) return 1;
  while (!qkh_reader_done(p)) {
    uint64_t total;
    uint32_t uni;
    int64_t n = qkh_reader_fill(p, seq, qual, off, cap_bytes, cap_reads, &total, &uni);
    if (n == -4) { /* a read longer than this (tiny) batch: retry the file with a bigger one */
      bad = -4;
      break;
    }
    if (n < 0) {
      bad = 1;
      break;
    }
    for (int64_t i = 0; i < n; i++) {
      const uint8_t *os, *oq;
      long l = oracle_reader_next(o, &os, &oq);
      uint64_t a = off[i], b = off[i + 1];
      if (l < 0 || (uint64_t)l != b - a || memcmp(os, seq + a, (size_t)l)) bad = 1;
      if (l >= 0 && oq && memcmp(oq, qual + a, (size_t)l)) bad = 1;
      if (l >= 0 && !oq)
        for (uint64_t k = a; k < b; k++)
          if (qual[k]) bad = 1;
    }
    if (bad) break;
  }
  if (!bad) {
    const uint8_t *os, *oq;
    if (oracle_reader_next(o, &os, &oq) >= 0) bad = 1; /* the product stopped early */
  }
  oracle_reader_close(o);
  qkh_reader_close(p);
  free(seq);
  free(qual);
  free(off);
  return bad;
}

int main(int argc, char **argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 300;
  rs = argc > 2 ? strtoull(argv[2], 0, 10) : 7;
  char *text = malloc(1 << 20);
  char path[64];
  snprintf(path, sizeof path, "/tmp/host_fuzz_%d.fq", (int)getpid());
  int failures = 0;
  for (int it = 0; it < iters; it++) {
    size_t n = make_text(text, 1 << 20);
    for (int gz = 0; gz < 3; gz++) {
      write_file(path, text, n, gz);
      /* ordinary gzip: every other round through the multi-threaded decoder,
       * with slices small enough for these files */
      if (gz == 1 && (it & 1)) {
        setenv("QUACK_PGZIP_CHUNK_KB", it % 4 == 1 ? "4" : "23", 1);
        setenv("QUACK_THREADS", it % 3 ? "3" : "7", 1);
      } else {
        unsetenv("QUACK_PGZIP_CHUNK_KB");
        setenv("QUACK_THREADS", "2", 1);
      }
      static const size_t caps[4] = {1 << 20, 4096, 1000, 257};
      for (int c = 0; c < 4; c++) {
        int rc = compare(path, caps[c], c == 2 ? 3 : 10000);
        if (rc == -4) continue;
        if (rc) {
          failures++;
          fprintf(stderr, "MISMATCH iter %d gz %d cap %zu (seed %llu)\n", it, gz, caps[c], (unsigned long long)rs);
          char keep[80];
          snprintf(keep, sizeof keep, "/tmp/host_fuzz_fail_%d_%d.fq", it, gz);
          write_file(keep, text, n, gz);
        }
      }
    }
  }
  unlink(path);
  /* large ordinary-gzip files: enough blocks for the speculative decoder of
   * pinflate.c to keep most of its slices */
  {
    const size_t big_cap = 5u << 20;
    char *big = malloc(big_cap);
    for (int it = 0; it < iters / 25 + 2; it++) {
      size_t n = 0;
      while (n + 200000 < big_cap) n += make_text(big + n, big_cap - n);
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      deflateInit2(&zs, 1 + (int)(rnd() % 9), Z_DEFLATED, 31, 1 + (int)(rnd() % 9), Z_DEFAULT_STRATEGY);
      size_t cap = deflateBound(&zs, n) + 64;
      unsigned char *out = malloc(cap);
      zs.next_in = (unsigned char *)big;
      zs.avail_in = (uInt)n;
      zs.next_out = out;
      zs.avail_out = (uInt)cap;
      deflate(&zs, Z_FINISH);
      FILE *f = fopen(path, "wb");
      fwrite(out, 1, cap - zs.avail_out, f);
      fclose(f);
      deflateEnd(&zs);
      free(out);
      setenv("QUACK_PGZIP_CHUNK_KB", it & 1 ? "4" : "64", 1);
      setenv("QUACK_THREADS", it % 3 ? "4" : "2", 1);
      if (compare(path, 1 << 20, 10000)) {
        failures++;
        printf("FAIL big it=%d n=%zu\n", it, n);
      }
    }
    free(big);
  }
  free(text);
  printf("host_fuzz: %d iterations, %d failures\n", iters, failures);
  return failures != 0;
}
