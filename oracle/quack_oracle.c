/*
 * quack_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see quack_oracle.h).
 *
 * A deliberately plain, single-threaded restatement of the reference's
 * accumulation path.  Each function cites the reference lines it follows.
 * The reference source itself cannot be compiled here (klib/kseq.h is an
 * un-vendored, unpinned submodule and is absent), so this file is pinned
 * against outputs of the reference's prebuilt binary instead
 * (oracle/make_goldens.sh, tests/test_oracle_pins.py).
 *
 * Behaviour outside the reference's defined domain (where quack.c indexes
 * arrays out of bounds) is fixed here, and identically in the HIP kernels:
 *   - quality byte b: bin = (b & 127) - 33; counted iff 0 <= bin <= 90
 *     (quack.c:203-204 would write past scores[91] otherwise);
 *   - base byte c: the letter key is (c & 31): 20 -> T(1), 3 -> C(2),
 *     7 -> G(3), everything else 0 — identical to lookup[(c-65)&~32]
 *     (quack.c:150,201) for every c in 'A'..'T' / 'a'..'t', total elsewhere;
 *   - zero-length reads only increment number_of_sequences (quack.c:219 would
 *     index bases[-1]).
 */
#include "quack_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ table */

void oracle_table_init(oracle_table *t) { memset(t, 0, sizeof *t); }

void oracle_table_free(oracle_table *t) {
  free(t->bases);
  memset(t, 0, sizeof *t);
}

/* quack.c:194-198: grow to the longest read seen, new tail zeroed */
static int table_reserve(oracle_table *t, uint64_t len) {
  if (len > t->capacity) {
    uint64_t cap = t->capacity ? t->capacity : 64;
    while (cap < len) cap *= 2;
    uint64_t *nb = realloc(t->bases, cap * ORACLE_ROWS * sizeof(uint64_t));
    if (!nb) return -1;
    memset(nb + t->capacity * ORACLE_ROWS, 0,
           (cap - t->capacity) * ORACLE_ROWS * sizeof(uint64_t));
    t->bases = nb;
    t->capacity = cap;
  }
  if (len > t->max_length) t->max_length = len;
  return 0;
}

/* -------------------------------------------------------------- bin rules */

int oracle_base_code(unsigned char c) {
  switch (c & 31u) {     /* quack.c:150: A C G T at lookup[0,2,6,19] */
    case 20: return 1;   /* T */
    case 3:  return 2;   /* C */
    case 7:  return 3;   /* G */
    default: return 0;   /* A and every other letter */
  }
}

int oracle_qual_bin(unsigned char b) {
  int bin = (int)(b & 127u) - 33;   /* quack.c:203 */
  return (bin >= 0 && bin <= 90) ? bin : -1;
}

/* ---------------------------------------------------------------- adapters */

/* quack.c:165-172: seed from s[0..9]; every later base rolls the 20-bit index
 * and marks it.  The seed window itself is never inserted. */
void oracle_adapter_insert(uint8_t *kmers, const char *seq, size_t len) {
  const uint32_t mask = ORACLE_KMER_TABLE - 1;
  uint32_t index = 0;
  size_t i;
  if (len <= 10) return;   /* the marking loop would not run (quack.c:169) */
  for (i = 0; i < 10; i++)
    index = ((index << 2) + (uint32_t)oracle_base_code((unsigned char)seq[i])) & mask;
  for (; i < len; i++) {
    index = ((index << 2) + (uint32_t)oracle_base_code((unsigned char)seq[i])) & mask;
    kmers[index] = 1;
  }
}

int oracle_read_adapters(const char *path, uint8_t *kmers) {
  oracle_reader *r = oracle_reader_open(path);
  const uint8_t *s, *q;
  long l;
  if (!r) return -1;
  memset(kmers, 0, ORACLE_KMER_TABLE);
  while ((l = oracle_reader_next(r, &s, &q)) >= 0)   /* quack.c:164 */
    oracle_adapter_insert(kmers, (const char *)s, (size_t)l);
  oracle_reader_close(r);
  return 0;
}

/* ------------------------------------------------------------ accumulation */

int oracle_accumulate_read(oracle_table *t, const uint8_t *seq,
                           const uint8_t *qual, size_t len,
                           const uint8_t *kmers) {
  const uint32_t mask = ORACLE_KMER_TABLE - 1;
  size_t i;
  if (table_reserve(t, len)) return -1;
  for (i = 0; i < len; i++) {                       /* quack.c:199-205 */
    uint64_t *b = t->bases + i * ORACLE_ROWS;
    int bin = oracle_qual_bin(qual[i]);
    b[91 + oracle_base_code(seq[i])]++;
    if (bin >= 0) b[bin]++;
  }
  /* quack.c:206-217.  The seed loop reads s[0..9] even when l < 10; the
   * result cannot matter then because i == 10 >= l. */
  i = 10;
  if (kmers && len >= 10) {
    uint32_t index = 0;
    size_t k;
    for (k = 0; k < 10; k++)
      index = ((index << 2) + (uint32_t)oracle_base_code(seq[k])) & mask;
    for (; kmers[index] == 0 && i < len; i++)
      index = ((index << 2) + (uint32_t)oracle_base_code(seq[i])) & mask;
  }
  if (i < len) t->bases[i * ORACLE_ROWS + 96]++;    /* kmer_count */
  if (len > 0) t->bases[(len - 1) * ORACLE_ROWS + 95]++;  /* quack.c:219 */
  t->number_of_sequences++;                          /* quack.c:220 */
  return 0;
}

int oracle_accumulate_batch(oracle_table *t, const uint8_t *seq,
                            const uint8_t *qual, const uint64_t *offsets,
                            uint64_t n_reads, uint32_t read_len,
                            const uint8_t *kmers) {
  uint64_t r;
  for (r = 0; r < n_reads; r++) {
    uint64_t start = offsets ? offsets[r] : r * (uint64_t)read_len;
    size_t len = offsets ? (size_t)(offsets[r + 1] - offsets[r]) : read_len;
    if (oracle_accumulate_read(t, seq + start, qual + start, len, kmers)) return -1;
  }
  return 0;
}

int oracle_read_fastq(const char *path, const uint8_t *kmers, oracle_table *t) {
  oracle_reader *r = oracle_reader_open(path);
  const uint8_t *s, *q;
  long l;
  if (!r) return -1;
  /* quack.c:193: the loop ends at the first negative return, EOF (-1) or a
   * malformed record (-2) alike. */
  while ((l = oracle_reader_next(r, &s, &q)) >= 0) {
    uint8_t *blank = NULL;
    int rc;
    if (!q) {  /* FASTA record fed to read_fastq: the reference reads a stale or
                  NULL qual.s (undefined); here no score is counted */
      blank = calloc((size_t)l + 1, 1);
      if (!blank) { oracle_reader_close(r); return -1; }
      q = blank;
    }
    rc = oracle_accumulate_read(t, s, q, (size_t)l, kmers);
    free(blank);
    if (rc) {
      oracle_reader_close(r);
      return -1;
    }
  }
  oracle_reader_close(r);
  return 0;
}

/* ------------------------------------------------------------- tokenizer
 * Restates the published kseq_read() of klib's kseq.h (MIT), the tokenizer
 * quack instantiates with KSEQ_INIT(gzFile, gzread) (quack.c:152):
 *   1. skip to the next '>' or '@' (unless the previous call already saw it);
 *   2. name = up to the first whitespace; the rest of the header line is a
 *      comment;
 *   3. sequence = every following line, concatenated, until a line starts
 *      with '>', '@' or '+'; empty lines are skipped; a trailing '\r' is
 *      dropped from a line when the accumulated length exceeds 1;
 *   4. FASTA record ends there.  For '+': skip that line, then append whole
 *      lines to the quality until it is at least as long as the sequence;
 *   5. EOF before any quality -> -2; lengths differ -> -2; EOF at a record
 *      boundary -> -1.
 */
typedef struct {
  uint8_t *s;
  size_t l, m;
} ostr;

struct oracle_reader {
  gzFile f;
  unsigned char buf[16384];
  int begin, end, eof;
  int last_char;
  ostr seq, qual;
};

static int rd_fill(oracle_reader *r) {
  if (r->eof) return 0;
  r->begin = 0;
  r->end = gzread(r->f, r->buf, sizeof r->buf);
  if (r->end <= 0) {
    r->end = 0;
    r->eof = 1;
    return 0;
  }
  return 1;
}

static int rd_getc(oracle_reader *r) {
  if (r->begin >= r->end && !rd_fill(r)) return -1;
  return r->buf[r->begin++];
}

static int ostr_push(ostr *s, const unsigned char *p, size_t n) {
  if (s->l + n + 1 > s->m) {
    size_t m = s->m ? s->m : 256;
    while (m < s->l + n + 1) m *= 2;
    uint8_t *ns = realloc(s->s, m);
    if (!ns) return -1;
    s->s = ns;
    s->m = m;
  }
  if (n) memcpy(s->s + s->l, p, n);
  s->l += n;
  s->s[s->l] = 0;
  return 0;
}

/* Append the rest of the current line to `dst` (or discard when dst is NULL).
 * Returns -1 when the stream was already exhausted, else 0.  *delim receives
 * the terminator ('\n') or 0 at EOF.  Mirrors ks_getuntil2(KS_SEP_LINE). */
static int rd_line(oracle_reader *r, ostr *dst, int *delim) {
  int got = 0;
  if (delim) *delim = 0;
  for (;;) {
    int i;
    if (r->begin >= r->end && !rd_fill(r)) break;
    for (i = r->begin; i < r->end; i++)
      if (r->buf[i] == '\n') break;
    got = 1;
    if (dst && ostr_push(dst, r->buf + r->begin, (size_t)(i - r->begin))) return -1;
    r->begin = i + 1;
    if (i < r->end) {
      if (delim) *delim = '\n';
      break;
    }
  }
  if (!got) return -1;
  if (dst && dst->l > 1 && dst->s[dst->l - 1] == '\r') dst->s[--dst->l] = 0;
  return 0;
}

/* header token: read up to the first whitespace; returns the separator or -1 */
static int rd_name(oracle_reader *r) {
  int got = 0;
  for (;;) {
    int i;
    if (r->begin >= r->end && !rd_fill(r)) break;
    got = 1;
    for (i = r->begin; i < r->end; i++) {
      unsigned char c = r->buf[i];
      if (c == ' ' || (c >= '\t' && c <= '\r')) {   /* isspace() in "C" locale */
        r->begin = i + 1;
        return c;
      }
    }
    r->begin = r->end;
  }
  return got ? 0 : -1;
}

oracle_reader *oracle_reader_open(const char *path) {
  oracle_reader *r = calloc(1, sizeof *r);
  if (!r) return NULL;
  r->f = gzopen(path, "r");
  if (!r->f) {
    free(r);
    return NULL;
  }
  return r;
}

void oracle_reader_close(oracle_reader *r) {
  if (!r) return;
  gzclose(r->f);
  free(r->seq.s);
  free(r->qual.s);
  free(r);
}

long oracle_reader_next(oracle_reader *r, const uint8_t **seq, const uint8_t **qual) {
  int c;
  if (r->last_char == 0) {
    while ((c = rd_getc(r)) >= 0 && c != '>' && c != '@') {}
    if (c < 0) return -1;
    r->last_char = c;
  }
  r->seq.l = r->qual.l = 0;
  if (ostr_push(&r->seq, NULL, 0) || ostr_push(&r->qual, NULL, 0)) return -3;
  c = rd_name(r);
  if (c < 0) return -1;                         /* EOF right after the marker */
  if (c != '\n' && c != 0) rd_line(r, NULL, NULL);  /* comment */
  while ((c = rd_getc(r)) >= 0 && c != '>' && c != '+' && c != '@') {
    unsigned char first = (unsigned char)c;
    if (c == '\n') continue;
    if (ostr_push(&r->seq, &first, 1)) return -3;
    if (rd_line(r, &r->seq, NULL) < 0) {
      /* stream ended right after `first`: kseq still applies the CR rule of
       * the (empty) append only when data was read, i.e. not here */
    }
  }
  if (c == '>' || c == '@') r->last_char = c;
  *seq = r->seq.s;
  *qual = NULL;
  if (c != '+') {
    if (c < 0) r->last_char = 0;
    return (long)r->seq.l;                      /* FASTA record */
  }
  while ((c = rd_getc(r)) >= 0 && c != '\n') {}  /* rest of the '+' line */
  if (c < 0) return -2;
  while (rd_line(r, &r->qual, NULL) >= 0 && r->qual.l < r->seq.l) {}
  r->last_char = 0;
  if (r->seq.l != r->qual.l) return -2;
  *qual = r->qual.s;
  return (long)r->seq.l;
}
