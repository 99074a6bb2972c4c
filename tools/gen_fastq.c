/*
 * gen_fastq — deterministic synthetic FASTQ generator (BASELINE.md §4 shapes).
 *
 *   gen_fastq OUT[.gz] N_READS LEN_LO LEN_HI SEED [Q_LO Q_HI [ADAPTERS.fa FRACTION]]
 *
 * Bases iid uniform over ACGT, qualities iid uniform integers in [Q_LO,Q_HI]
 * (Phred+33), headers "@r<i>", bare "+" line, LF endings.  With an adapter
 * FASTA, FRACTION of the reads get one adapter spliced in at a uniform offset
 * (truncated at the read end).  OUT ending in .gz is written through zlib
 * (level 6), OUT ending in .bgz as BGZF (independent <= 64 KiB gzip members
 * with a "BC" size field, what bgzip writes), "-" is stdout.  PRNG: splitmix64.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static uint64_t state;
static uint64_t next64(void) {
  uint64_t z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint32_t below(uint32_t n) { return (uint32_t)((next64() >> 32) * (uint64_t)n >> 32); }

/* one BGZF member holding `n` (<= 65280) bytes */
static void bgzf_block(FILE *f, const unsigned char *data, size_t n) {
  unsigned char out[70000];
  z_stream s;
  memset(&s, 0, sizeof s);
  deflateInit2(&s, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
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

int main(int argc, char **argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s OUT[.gz] N LEN_LO LEN_HI SEED [Q_LO Q_HI [ADAPTERS.fa FRACTION]]\n", argv[0]);
    return 2;
  }
  const char *out = argv[1];
  uint64_t n = strtoull(argv[2], 0, 10);
  uint32_t lo = (uint32_t)atoi(argv[3]), hi = (uint32_t)atoi(argv[4]);
  state = strtoull(argv[5], 0, 10);
  int qlo = argc > 7 ? atoi(argv[6]) : 2, qhi = argc > 7 ? atoi(argv[7]) : 41;
  char **ads = NULL;
  size_t n_ads = 0;
  double frac = 0;
  if (argc > 9) {
    FILE *f = fopen(argv[8], "r");
    char line[4096];
    if (!f) { perror(argv[8]); return 1; }
    while (fgets(line, sizeof line, f)) {
      if (line[0] == '>' || line[0] == '\n') continue;
      line[strcspn(line, "\r\n")] = 0;
      ads = realloc(ads, (n_ads + 1) * sizeof *ads);
      ads[n_ads++] = strdup(line);
    }
    fclose(f);
    frac = atof(argv[9]);
  }
  size_t ln = strlen(out);
  int gz = ln > 3 && !strcmp(out + ln - 3, ".gz");
  int bgz = ln > 4 && !strcmp(out + ln - 4, ".bgz");
  gzFile g = NULL;
  FILE *fp = NULL;
  unsigned char *pend = bgz ? malloc(65280 + 2 * (size_t)hi + 64) : NULL;
  size_t n_pend = 0;
  if (gz) { g = gzopen(out, "wb6"); gzbuffer(g, 1 << 20); }
  else fp = strcmp(out, "-") ? fopen(out, "wb") : stdout;
  if (!g && !fp) { perror(out); return 1; }
  char *buf = malloc(2 * (size_t)hi + 64);
  for (uint64_t i = 0; i < n; i++) {
    uint32_t l = lo + (hi > lo ? below(hi - lo + 1) : 0);
    int k = sprintf(buf, "@r%llu\n", (unsigned long long)i);
    char *s = buf + k;
    for (uint32_t j = 0; j < l; j += 32) {
      uint64_t r = next64();
      for (uint32_t t = 0; t < 32 && j + t < l; t++) s[j + t] = "ACGT"[(r >> (2 * t)) & 3];
    }
    if (n_ads && (double)(next64() >> 11) / 9007199254740992.0 < frac) {
      const char *a = ads[below((uint32_t)n_ads)];
      uint32_t at = below(l ? l : 1), al = (uint32_t)strlen(a);
      if (al > l - at) al = l - at;
      memcpy(s + at, a, al);
    }
    char *q = s + l;
    *q++ = '\n'; *q++ = '+'; *q++ = '\n';
    for (uint32_t j = 0; j < l; j++) q[j] = (char)(33 + qlo + (int)below((uint32_t)(qhi - qlo + 1)));
    q[l] = '\n';
    size_t tot = (size_t)(q + l + 1 - buf);
    if (gz) gzwrite(g, buf, (unsigned)tot);
    else if (bgz) {
      size_t off = 0;
      while (off < tot) {
        size_t k = tot - off < 65280 - n_pend ? tot - off : 65280 - n_pend;
        memcpy(pend + n_pend, buf + off, k);
        n_pend += k;
        off += k;
        if (n_pend == 65280) { bgzf_block(fp, pend, n_pend); n_pend = 0; }
      }
    } else fwrite(buf, 1, tot, fp);
  }
  if (bgz) {
    if (n_pend) bgzf_block(fp, pend, n_pend);
    bgzf_block(fp, pend, 0);   /* end-of-file marker */
  }
  if (gz) gzclose(g); else if (fp != stdout) fclose(fp);
  return 0;
}
