/*
 * reader.c — FASTQ/FASTA tokenizer + batcher feeding the accumulation path.
 *
 * Same record grammar as klib's kseq_read(), which quack instantiates over
 * gzread (quack.c:152, 158-175, 182-222; klib is an absent submodule, so the
 * grammar is implemented from its published behaviour and pinned by the
 * reference binary's outputs in tests/):
 *   - a record starts at the next '>' or '@'; name = up to first whitespace,
 *     the rest of that line is ignored;
 *   - sequence lines are concatenated until a line starts with '>', '@' or
 *     '+'; empty lines are skipped; one trailing CR is dropped per line once
 *     the accumulated text is longer than one byte;
 *   - after '+', whole lines are appended to the quality until it is at least
 *     as long as the sequence; a length mismatch or a missing quality ends the
 *     stream (the reference's read loop stops at the first negative return,
 *     quack.c:193).
 * Input bytes come in blocks from source.c (parallel / threaded inflate).
 * Unlike kseq, bytes go straight into the caller's batch arrays (the pinned
 * staging buffers of the C-ABI): one memchr + one memcpy per line.  A record
 * that does not fit in the rest of the batch is parked on the heap and opens
 * the next batch.
 */
#include <stdlib.h>
#include <string.h>

#include "quack_host.h"
#include "source.h"

/* where the bytes of the record being parsed go */
typedef struct {
  uint8_t *dst;        /* current destination (batch memory or heap) */
  size_t len, room;    /* bytes written / capacity of dst */
  uint8_t **heap;      /* parking buffer (owned by the reader) */
  size_t *heap_cap;
  int parked;          /* 1: dst == *heap */
} sink;

struct qkh_reader {
  qkh_source *src;
  const uint8_t *buf;  /* the source block being parsed */
  size_t pos, lim;
  const uint32_t *nl;  /* the block's newline offsets, indexed by the producer threads (NULL: none) */
  size_t n_nl, li;     /* ... how many, and the first one not known to lie before pos */
  int have_block;
  int eof;
  int marker;          /* header marker already consumed ('>' / '@'), or 0 */
  int finished;        /* no further records will be produced */
  uint8_t *park_seq, *park_qual;
  size_t park_seq_cap, park_qual_cap;
  size_t park_len;
  int have_parked, parked_is_fastq;
};

static int refill(qkh_reader *r) {
  if (r->eof) return 0;
  r->pos = 0;
  r->nl = NULL;
  r->n_nl = r->li = 0;
  if (!qkh_source_next(r->src, &r->buf, &r->lim)) {
    r->lim = 0;
    r->eof = 1;
    r->have_block = 0;
    return 0;
  }
  if (!qkh_source_lines(r->src, &r->nl, &r->n_nl)) r->nl = NULL;
  r->have_block = 1;
  return 1;
}

static inline int next_byte(qkh_reader *r) {
  if (r->pos >= r->lim && !refill(r)) return -1;
  return r->buf[r->pos++];
}

/* move the record's bytes written so far to the reader-owned heap buffer */
static int sink_park(sink *s, size_t need) {
  size_t cap = *s->heap_cap ? *s->heap_cap : 4096;
  while (cap < need) cap *= 2;
  if (cap > *s->heap_cap || !*s->heap) {
    uint8_t *h = realloc(*s->heap, cap);
    if (!h) return -1;
    *s->heap = h;
    *s->heap_cap = cap;
  }
  if (!s->parked && s->len) memcpy(*s->heap, s->dst, s->len);
  s->dst = *s->heap;
  s->room = *s->heap_cap;
  s->parked = 1;
  return 0;
}

static int sink_put(sink *s, const uint8_t *p, size_t n) {
  if (s->len + n > s->room && sink_park(s, s->len + n)) return -1;
  if (n) memcpy(s->dst + s->len, p, n);
  s->len += n;
  return 0;
}

/* Consume through the next '\n'.  With a sink, the line's bytes are appended
 * and the CR rule applied.  Returns 1 if any input was available, 0 if the
 * stream was already exhausted, -1 on allocation failure. */
static int take_line(qkh_reader *r, sink *s) {
  int any = 0;
  for (;;) {
    const uint8_t *from, *nl;
    size_t avail;
    if (r->pos >= r->lim && !refill(r)) break;
    any = 1;
    from = r->buf + r->pos;
    avail = r->lim - r->pos;
    nl = memchr(from, '\n', avail);
    if (nl) {
      if (s && sink_put(s, from, (size_t)(nl - from))) return -1;
      r->pos += (size_t)(nl - from) + 1;
      break;
    }
    if (s && sink_put(s, from, avail)) return -1;
    r->pos = r->lim;
  }
  if (any && s && s->len > 1 && s->dst[s->len - 1] == '\r') s->len--;
  return any;
}

/* Skip the record name; returns the whitespace byte that ended it, 0 when the
 * stream ended inside the name, -1 when there was nothing to read. */
static int skip_name(qkh_reader *r) {
  int any = 0;
  for (;;) {
    if (r->pos >= r->lim && !refill(r)) return any ? 0 : -1;
    any = 1;
    while (r->pos < r->lim) {
      uint8_t c = r->buf[r->pos++];
      if (c == ' ' || (c >= 9 && c <= 13)) return c;
    }
  }
}

/* One record.  Returns its length, -1 at end of stream, -2 for a malformed
 * record, -3 when out of memory. */
static long parse_record(qkh_reader *r, sink *sq, sink *ql, int *is_fastq) {
  int c;
  if (!r->marker) {
    do c = next_byte(r); while (c >= 0 && c != '>' && c != '@');
    if (c < 0) return -1;
    r->marker = c;
  }
  /* Header line.  The name ends at the first whitespace; if that is not the
   * newline the rest of the line is skipped as well — either way the parser
   * ends up just behind the first '\n', so when one is in sight a memchr does
   * the whole line (Illumina headers are 40-70 bytes: the per-byte scan of
   * skip_name was a fifth of the tokenizer's time).  The careful path below
   * only sees lines cut by a block boundary or by the end of the stream. */
  {
    const uint8_t *nl = r->pos < r->lim ? memchr(r->buf + r->pos, '\n', r->lim - r->pos) : NULL;
    if (nl) {
      r->pos = (size_t)(nl - r->buf) + 1;
    } else {
      c = skip_name(r);
      if (c < 0) return -1;
      if (c != '\n' && c != 0 && take_line(r, NULL) < 0) return -3;
    }
  }
  for (;;) {
    uint8_t first;
    c = next_byte(r);
    if (c < 0 || c == '>' || c == '+' || c == '@') break;
    if (c == '\n') continue;
    first = (uint8_t)c;
    if (sink_put(sq, &first, 1) || take_line(r, sq) < 0) return -3;
  }
  r->marker = (c == '>' || c == '@') ? c : 0;
  *is_fastq = c == '+';
  if (c != '+') {
    if (c >= 0) return (long)sq->len;
    r->marker = 0;
    return (long)sq->len;
  }
  {  /* rest of the '+' line */
    const uint8_t *nl = r->pos < r->lim ? memchr(r->buf + r->pos, '\n', r->lim - r->pos) : NULL;
    if (nl) {
      r->pos = (size_t)(nl - r->buf) + 1;
    } else {
      do c = next_byte(r); while (c >= 0 && c != '\n');
      if (c < 0) return -2;
    }
  }
  for (;;) {
    int got = take_line(r, ql);
    if (got < 0) return -3;
    if (got == 0 || ql->len >= sq->len) break;
  }
  r->marker = 0;
  return ql->len == sq->len ? (long)sq->len : -2;
}

qkh_reader *qkh_reader_open(const char *path) {
  qkh_reader *r = calloc(1, sizeof *r);
  if (!r) return NULL;
  r->src = qkh_source_open(path);
  if (!r->src) {
    free(r);
    return NULL;
  }
  return r;
}

void qkh_reader_close(qkh_reader *r) {
  if (!r) return;
  qkh_source_close(r->src);
  free(r->park_seq);
  free(r->park_qual);
  free(r);
}

int qkh_reader_failed(const qkh_reader *r) { return qkh_source_failed(r->src); }

int qkh_reader_done(const qkh_reader *r) { return r->finished && !r->have_parked; }

/* The common case without the general machinery: a four-line FASTQ record
 * (header / one sequence line / '+' line / one quality line of the same
 * length, no CR) that lies completely inside the current block and fits the
 * batch.  Copies it and returns its length, or returns 0 having consumed
 * nothing — every other shape (multi-line, FASTA, CRLF, empty lines, records
 * cut by a block boundary, parked records, the end of the stream) is left to
 * parse_record, which gives the same answer for these. */
static size_t fast_record(qkh_reader *r, uint8_t *seq_dst, uint8_t *qual_dst, size_t room) {
  const uint8_t *p, *end, *h, *s0, *e1, *e2, *q0;
  size_t len;
  if (!r->have_block || r->pos >= r->lim) return 0;
  p = r->buf + r->pos;
  end = r->buf + r->lim;
  if (r->nl) {
    /* The same record shape and the same tests, with the four line ends read off the producers'
     * index instead of four memchr calls (the tokenizer is one thread and paces the whole file). */
    const uint32_t *nl = r->nl;
    size_t li = r->li;
    while (li < r->n_nl && nl[li] < r->pos) li++;   /* (after the general parser consumed lines) */
    r->li = li;
    if (li + 4 > r->n_nl) return 0;
    if (r->marker) {
      if (r->marker != '@') return 0;
    } else if (*p != '@') {
      return 0;
    }
    h = r->buf + nl[li];
    s0 = h + 1;
    e1 = r->buf + nl[li + 1];
    e2 = r->buf + nl[li + 2];
    q0 = e2 + 1;
    len = (size_t)(e1 - s0);
    if (len == 0 || *s0 == '>' || *s0 == '+' || *s0 == '@') return 0;
    if (e1[-1] == '\r' || e1[1] != '+') return 0;
    if (r->buf + nl[li + 3] != q0 + len || q0[len - 1] == '\r') return 0;
    if (len > room) return 0;
    memcpy(seq_dst, s0, len);
    memcpy(qual_dst, q0, len);
    r->pos = (size_t)nl[li + 3] + 1;
    r->li = li + 4;
    r->marker = 0;
    return len;
  }
  if (r->marker) {
    if (r->marker != '@') return 0;
  } else {
    if (*p != '@') return 0;
    p++;
  }
  if (!(h = memchr(p, '\n', (size_t)(end - p)))) return 0;
  s0 = h + 1;
  if (s0 >= end || *s0 == '>' || *s0 == '+' || *s0 == '@' || *s0 == '\n') return 0;
  if (!(e1 = memchr(s0, '\n', (size_t)(end - s0)))) return 0;
  len = (size_t)(e1 - s0);
  if (e1[-1] == '\r' || e1 + 1 >= end || e1[1] != '+') return 0;
  if (!(e2 = memchr(e1 + 1, '\n', (size_t)(end - (e1 + 1))))) return 0;
  q0 = e2 + 1;
  if ((size_t)(end - q0) < len + 1 || q0[len] != '\n' || q0[len - 1] == '\r' || memchr(q0, '\n', len)) return 0;
  if (len > room) return 0;
  memcpy(seq_dst, s0, len);
  memcpy(qual_dst, q0, len);
  r->pos = (size_t)(q0 - r->buf) + len + 1;
  r->marker = 0;
  return len;
}

/* One batch.  Read i is written at starts[i], the end of its predecessor
 * rounded up to `align`; lengths may be NULL (packed batches: the caller reads
 * the ends off the next start).  stride != 0: read i is written at i * stride
 * instead (starts may be NULL), and the batch ends in front of the first read
 * that is longer than the stride — it is parked for the next batch. */
static int64_t fill_batch(qkh_reader *r, uint8_t *seq, uint8_t *qual, uint64_t *starts, uint32_t *lengths,
                          uint64_t cap_bytes, uint64_t cap_reads, uint64_t align, uint64_t *extent,
                          uint32_t *uniform_len, uint64_t stride) {
  uint64_t n = 0, total = 0;   /* total = end of the last read */
  int64_t common = -1;         /* -1 unknown, -2 mixed */
  /* strided batches: the bytes behind a read's last base, up to the stride, are set to 0xFF in both arrays — neutral to the
   * kernels (QK_BATCH_NEUTRAL_PADS in quack_hip.h), which then need no tail masks */
#define PAD_FF(at_, len_)                                                         \
  do {                                                                            \
    if (stride && (uint64_t)(len_) < stride) {                                    \
      memset(seq + (at_) + (len_), 0xFF, (size_t)(stride - (uint64_t)(len_)));    \
      memset(qual + (at_) + (len_), 0xFF, (size_t)(stride - (uint64_t)(len_)));   \
    }                                                                             \
  } while (0)
  if (stride && cap_reads > cap_bytes / stride) cap_reads = cap_bytes / stride;
  if (r->have_parked) {
    if (stride && r->park_len > stride) {     /* the caller has to pick another layout first */
      *extent = 0;
      *uniform_len = 0;
      return 0;
    }
    if (r->park_len > cap_bytes) return -4;   /* a single read exceeds the batch */
    memcpy(seq, r->park_seq, r->park_len);
    if (r->parked_is_fastq) memcpy(qual, r->park_qual, r->park_len);
    else memset(qual, 0, r->park_len);
    if (starts) starts[0] = 0;
    if (lengths) lengths[0] = (uint32_t)r->park_len;
    PAD_FF(0, r->park_len);
    total = r->park_len;
    n = 1;
    common = (int64_t)r->park_len;
    r->have_parked = 0;
  }
  while (!r->finished && n < cap_reads) {
    const uint64_t at = stride ? n * stride : (total + align - 1) & ~(align - 1);
    const uint64_t room = stride ? stride : cap_bytes - at;
    if (at >= cap_bytes && n > 0) break;      /* no room left for another start */
    const size_t fl = fast_record(r, seq + at, qual + at, room);
    if (fl) {
      if (fl > 0xFFFFFFFFull) return -4;
      if (starts) starts[n] = at;
      if (lengths) lengths[n] = (uint32_t)fl;
      PAD_FF(at, fl);
      n++;
      total = at + fl;
      if (common == -1) common = (int64_t)fl;
      else if (common != (int64_t)fl) common = -2;
      continue;
    }
    sink sq = {seq + at, 0, room, &r->park_seq, &r->park_seq_cap, 0};
    sink ql = {qual + at, 0, room, &r->park_qual, &r->park_qual_cap, 0};
    int is_fastq = 0;
    long l = parse_record(r, &sq, &ql, &is_fastq);
    if (l == -3) return -3;
    if (l < 0) {
      r->finished = 1;
      break;
    }
    if (sq.parked || ql.parked) {
      /* did not fit: keep the whole record on the heap for the next batch */
      if (!sq.parked && sink_park(&sq, sq.len + 1)) return -3;
      if (!ql.parked && sink_park(&ql, ql.len + 1)) return -3;
      r->park_len = (size_t)l;
      r->have_parked = 1;
      r->parked_is_fastq = is_fastq;
      if (n == 0 && !stride && (uint64_t)l > cap_bytes) return -4;
      break;
    }
    if (!is_fastq) memset(qual + at, 0, (size_t)l);  /* FASTA fed as reads: no scores */
    if (starts) starts[n] = at;
    if (lengths) {
      if ((uint64_t)l > 0xFFFFFFFFull) return -4;
      lengths[n] = (uint32_t)l;
    }
    PAD_FF(at, l);
    n++;
    total = at + (uint64_t)l;
    if (common == -1) common = l;
    else if (common != l) common = -2;
  }
#undef PAD_FF
  *extent = total;
  *uniform_len = (common > 0 && common <= 0x7FFFFFFF) ? (uint32_t)common : 0;
  return (int64_t)n;
}

int64_t qkh_reader_fill(qkh_reader *r, uint8_t *seq, uint8_t *qual,
                        uint64_t *offsets, uint64_t cap_bytes,
                        uint64_t cap_reads, uint64_t *total_bytes,
                        uint32_t *uniform_len) {
  int64_t n;
  offsets[0] = 0;
  n = fill_batch(r, seq, qual, offsets, NULL, cap_bytes, cap_reads, 1, total_bytes, uniform_len, 0);
  if (n >= 0) offsets[n] = *total_bytes;   /* packed: read i ends where i+1 starts */
  return n;
}

int64_t qkh_reader_fill_gapped(qkh_reader *r, uint8_t *seq, uint8_t *qual,
                               uint64_t *starts, uint32_t *lengths, uint64_t cap_bytes,
                               uint64_t cap_reads, uint64_t align,
                               uint64_t *extent_bytes, uint32_t *uniform_len) {
  if (!align || (align & (align - 1)) || !lengths) return -3;
  return fill_batch(r, seq, qual, starts, lengths, cap_bytes, cap_reads, align, extent_bytes, uniform_len, 0);
}

int64_t qkh_reader_fill_strided(qkh_reader *r, uint8_t *seq, uint8_t *qual, uint32_t *lengths,
                                uint64_t cap_bytes, uint64_t cap_reads, uint32_t stride, uint32_t *uniform_len) {
  uint64_t extent;
  if (!stride || !lengths) return -3;
  return fill_batch(r, seq, qual, NULL, lengths, cap_bytes, cap_reads, 1, &extent, uniform_len, stride);
}

uint64_t qkh_reader_parked_len(const qkh_reader *r) { return r->have_parked ? (uint64_t)r->park_len : 0; }

/* ----------------------------------------------------------------- adapters */

static inline uint32_t letter_code(uint8_t c) {
  /* quack.c:148-150: A C G T at lookup[0], [2], [6], [19]; (c-65)&~32 == key-1
   * for letters, and key = c & 31 */
  switch (c & 31u) {
    case 20: return 1;
    case 3: return 2;
    case 7: return 3;
    default: return 0;
  }
}

void qkh_adapter_insert(uint32_t *bitset, const uint8_t *seq, uint64_t len) {
  uint32_t idx = 0;
  uint64_t i;
  if (len <= QK_KMER_SIZE) return;
  for (i = 0; i < QK_KMER_SIZE; i++)                 /* quack.c:166-168 */
    idx = ((idx << 2) + letter_code(seq[i])) & (QK_KMER_TABLE_BITS - 1);
  for (; i < len; i++) {                             /* quack.c:169-172 */
    idx = ((idx << 2) + letter_code(seq[i])) & (QK_KMER_TABLE_BITS - 1);
    bitset[idx >> 5] |= 1u << (idx & 31);
  }
}

int qkh_read_adapters(const char *path, uint32_t *bitset) {
  qkh_reader *r = qkh_reader_open(path);
  uint8_t *s = NULL, *q = NULL;
  size_t scap = 0, qcap = 0;
  if (!r) return -1;
  memset(bitset, 0, QK_KMER_TABLE_WORDS * sizeof(uint32_t));
  for (;;) {
    sink sq = {NULL, 0, 0, &s, &scap, 0};
    sink ql = {NULL, 0, 0, &q, &qcap, 0};
    int is_fastq;
    long l = parse_record(r, &sq, &ql, &is_fastq);
    if (l < 0) break;                                /* quack.c:164 */
    qkh_adapter_insert(bitset, sq.dst, (uint64_t)l);
  }
  free(s);
  free(q);
  const int failed = qkh_reader_failed(r);   /* a decoder thread out of memory: not the file's end */
  qkh_reader_close(r);
  return failed ? -1 : 0;
}
