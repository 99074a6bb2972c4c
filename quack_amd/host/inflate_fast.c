/*
 * inflate_fast.c — gzip / DEFLATE (RFC 1951, 1952) decoder for the host feed.
 *
 * Why: the reference spends 78 % of its wall time in zlib's inflate (SURVEY
 * 3.3) and, with the accumulation on the GPU, inflate is the whole end-to-end
 * limiter.  This decoder follows the usual fast-inflate recipe: a 64-bit bit
 * buffer refilled with one unaligned 8-byte load, multi-level decode tables
 * whose entries carry base value + extra-bit count + code length, a fast loop
 * that runs while both buffers have slack, and 8-bytes-at-a-time match copies.
 * It decodes from a memory-mapped file straight into the reader's blocks and is
 * resumable at any output position (a match may be cut by the end of a block).
 *
 * Scope: gzip members (any number, concatenated), all three block types.
 * Member trailers (CRC-32, ISIZE) are logged for the caller, which owns the
 * bytes of a member even when several decoders produced them: source.c chains
 * the CRCs of the pieces and ends the stream where zlib's gzread — the
 * reference's reader, quack.c:187,193 — would.  Anything else this decoder
 * rejects makes the caller stop exactly there, like a failing gzread.
 * tests/test_inflate.py fuzzes it against zlib.
 */
#include "inflate_fast.h"

#include <stdlib.h>
#include <string.h>

#define LITLEN_BITS 11
#define DIST_BITS 8
#define MULTI_BITS QKH_MULTI_BITS
/* the fast loop takes four literal-run lookups from one refill of >= 56 bits, and a table entry's "bits to consume" field is 6 bits wide */
_Static_assert(4 * QKH_MULTI_BITS <= 56 && QKH_MULTI_BITS <= 14 && QKH_MULTI_BITS >= 9, "QKH_MULTI_BITS: four lookups must fit one 56-bit refill");
#define MAX_CODE_LEN 15

/* table entry: bits 0-4 code length to consume, 5-7 kind, 8-12 extra bits
 * (or sub-table index width), 16-31 value (literal / base / sub-table offset) */
enum { K_LIT = 0, K_BASE = 1, K_END = 2, K_SUB = 3, K_BAD = 7 };
#define ENTRY(val, extra, kind, len) (((uint32_t)(val) << 16) | ((uint32_t)(extra) << 8) | ((uint32_t)(kind) << 5) | (uint32_t)(len))
#define E_LEN(e) ((e) & 31u)
#define E_KIND(e) (((e) >> 5) & 7u)
#define E_EXTRA(e) (((e) >> 8) & 31u)
#define E_VAL(e) ((e) >> 16)

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static uint32_t rev_bits(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i);
  return r;
}

/* Build a (primary + sub-tables) decode table from code lengths.  `kind_of`
 * maps a symbol to its entry.  Returns 0, or -1 for an over-subscribed or
 * (non-trivially) incomplete code. */
static int build_table(uint32_t *table, int table_cap, int primary_bits, const uint8_t *lens, int n_syms,
                       uint32_t (*entry_of)(int sym, int len)) {
  int count[MAX_CODE_LEN + 1] = {0};
  uint32_t next_code[MAX_CODE_LEN + 2];
  int max_len = 0, used = 1 << primary_bits;
  for (int i = 0; i < n_syms; i++) count[lens[i]]++;
  count[0] = 0;
  for (int l = MAX_CODE_LEN; l > 0; l--)
    if (count[l]) {
      max_len = l;
      break;
    }
  for (int i = 0; i < (1 << primary_bits); i++) table[i] = ENTRY(0, 0, K_BAD, 0);
  if (max_len == 0) return 0; /* no codes: every lookup is invalid */
  {
    uint32_t code = 0;
    long left = 1;
    for (int l = 1; l <= MAX_CODE_LEN; l++) {
      left <<= 1;
      left -= count[l];
      if (left < 0) return -1; /* over-subscribed */
      code = (code + (uint32_t)count[l - 1]) << 1;
      next_code[l] = code;
    }
    /* an incomplete code is only legal when it is a single 1-bit code
     * (RFC 1951 3.2.7; zlib enforces the same) */
    if (left > 0 && max_len != 1) return -1;
  }
  /* sub-table sizing: for every primary prefix of a long code, the widest
   * remaining length decides the sub-table's index width */
  if (max_len > primary_bits) {
    /* first pass: width per prefix */
    uint8_t width[1 << LITLEN_BITS];   /* on the stack: two files may be decoded concurrently */
    uint32_t nc[MAX_CODE_LEN + 2];
    memcpy(nc, next_code, sizeof nc);
    memset(width, 0, (size_t)1 << primary_bits);
    for (int s = 0; s < n_syms; s++) {
      int l = lens[s];
      if (l <= primary_bits) {
        if (l) nc[l]++;
        continue;
      }
      uint32_t r = rev_bits(nc[l]++, l);
      uint32_t prefix = r & ((1u << primary_bits) - 1u);
      if (l - primary_bits > width[prefix]) width[prefix] = (uint8_t)(l - primary_bits);
    }
    for (int pfx = 0; pfx < (1 << primary_bits); pfx++)
      if (width[pfx]) {
        if (used + (1 << width[pfx]) > table_cap) return -1;
        table[pfx] = ENTRY(used, width[pfx], K_SUB, primary_bits);
        for (int i = 0; i < (1 << width[pfx]); i++) table[used + i] = ENTRY(0, 0, K_BAD, 0);
        used += 1 << width[pfx];
      }
  }
  for (int s = 0; s < n_syms; s++) {
    int l = lens[s];
    if (!l) continue;
    uint32_t r = rev_bits(next_code[l]++, l);
    if (l <= primary_bits) {
      uint32_t e = entry_of(s, l);
      for (uint32_t i = r; i < (1u << primary_bits); i += 1u << l) table[i] = e;
    } else {
      uint32_t prefix = r & ((1u << primary_bits) - 1u);
      uint32_t sub = table[prefix];
      int w = (int)E_EXTRA(sub), rest = l - primary_bits;
      uint32_t e = entry_of(s, rest);
      for (uint32_t i = r >> primary_bits; i < (1u << w); i += 1u << rest) table[E_VAL(sub) + i] = e;
    }
  }
  return 0;
}

static uint32_t litlen_entry(int sym, int len) {
  if (sym < 256) return ENTRY(sym, 0, K_LIT, len);
  if (sym == 256) return ENTRY(0, 0, K_END, len);
  if (sym > 285) return ENTRY(0, 0, K_BAD, len);
  return ENTRY(len_base[sym - 257], len_extra[sym - 257], K_BASE, len);
}
static uint32_t dist_entry(int sym, int len) {
  if (sym > 29) return ENTRY(0, 0, K_BAD, len);
  return ENTRY(dist_base[sym], dist_extra[sym], K_BASE, len);
}
static uint32_t plain_entry(int sym, int len) { return ENTRY(sym, 0, K_LIT, len); }

/* ------------------------------------------------------------ bit reader */
#define NEED_INPUT_SLACK 16

static inline uint64_t load64(const uint8_t *p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}

/* careful refill: byte by byte, zeros past the end (over-reads are caught by
 * comparing consumed bits with the input size) */
static inline void refill_slow(qkh_inflate *z) {
  while (z->bitcnt < 56) {   /* ends with 56..63 valid bits: the fast refill's shift stays < 64 */
    uint64_t b = z->in < z->in_end ? *z->in : 0;
    z->in++;
    z->bitbuf |= b << z->bitcnt;
    z->bitcnt += 8;
  }
}
static inline int overran(const qkh_inflate *z) {
  /* bytes actually consumed = in - bitcnt/8 must not pass in_end */
  return z->in - (z->bitcnt >> 3) > z->in_end;
}
#define PEEK(z, n) ((uint32_t)((z)->bitbuf & ((1ull << (n)) - 1ull)))
#define DROP(z, n) ((z)->bitbuf >>= (n), (z)->bitcnt -= (int)(n))

static uint32_t take_bits(qkh_inflate *z, int n) {
  uint32_t v;
  if (z->bitcnt < n) refill_slow(z);
  v = PEEK(z, n);
  DROP(z, n);
  return v;
}

static inline uint32_t decode_sym(qkh_inflate *z, const uint32_t *table, int primary_bits) {
  uint32_t e = table[PEEK(z, primary_bits)];
  if (E_KIND(e) == K_SUB) {
    DROP(z, primary_bits);
    e = table[E_VAL(e) + PEEK(z, E_EXTRA(e))];
  }
  DROP(z, E_LEN(e));
  return e;
}

/* ------------------------------------------------------------ block headers */
static int read_gzip_header(qkh_inflate *z) {
  /* byte-aligned here */
  const uint8_t *p = z->in, *e = z->in_end;
  if (e - p < 18) return -1;
  if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return -1;
  int flg = p[3];
  p += 10;
  if (flg & 4) { /* FEXTRA */
    if (e - p < 2) return -1;
    size_t xl = p[0] | ((size_t)p[1] << 8);
    p += 2;
    if ((size_t)(e - p) < xl) return -1;
    p += xl;
  }
  if (flg & 8) { /* FNAME */
    while (p < e && *p) p++;
    if (p == e) return -1;
    p++;
  }
  if (flg & 16) { /* FCOMMENT */
    while (p < e && *p) p++;
    if (p == e) return -1;
    p++;
  }
  if (flg & 2) { /* FHCRC */
    if (e - p < 2) return -1;
    p += 2;
  }
  z->in = p;
  z->member_out = 0;
  return 0;
}

static void align_to_byte(qkh_inflate *z) {
  /* give whole unread bytes back to the input pointer */
  DROP(z, z->bitcnt & 7);
  z->in -= z->bitcnt >> 3;
  z->bitbuf = 0;
  z->bitcnt = 0;
}

int qkh_read_block_header(qkh_inflate *z) {
  refill_slow(z);
  z->final_block = (int)take_bits(z, 1);
  uint32_t type = take_bits(z, 2);
  if (type == 0) {
    align_to_byte(z);
    if (z->in > z->in_end || z->in_end - z->in < 4) return -1;
    uint32_t len = z->in[0] | ((uint32_t)z->in[1] << 8), nlen = z->in[2] | ((uint32_t)z->in[3] << 8);
    if ((len ^ nlen) != 0xFFFFu) return -1;
    z->in += 4;
    z->stored_left = len;
    z->state = QKH_Z_STORED;
    return 0;
  }
  if (type == 1) {
    if (!z->fixed_ready) {
      uint8_t l[288];
      int i = 0;
      for (; i < 144; i++) l[i] = 8;
      for (; i < 256; i++) l[i] = 9;
      for (; i < 280; i++) l[i] = 7;
      for (; i < 288; i++) l[i] = 8;
      if (build_table(z->fixed_litlen, QKH_LITLEN_TABLE, LITLEN_BITS, l, 288, litlen_entry)) return -1;
      for (i = 0; i < 32; i++) l[i] = 5;
      if (build_table(z->fixed_dist, QKH_DIST_TABLE, DIST_BITS, l, 32, dist_entry)) return -1;
      z->fixed_ready = 1;
    }
    z->litlen = z->fixed_litlen;
    z->dist = z->fixed_dist;
    z->multi = NULL;
    z->state = QKH_Z_CODES;
    return 0;
  }
  if (type == 2) {
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0}, lens[288 + 32];
    uint32_t cl_table[1 << 7];
    refill_slow(z);
    int hlit = (int)take_bits(z, 5) + 257, hdist = (int)take_bits(z, 5) + 1, hclen = (int)take_bits(z, 4) + 4;
    if (hlit > 286 || hdist > 30) return -1;
    for (int i = 0; i < hclen; i++) cl[order[i]] = (uint8_t)take_bits(z, 3);
    if (build_table(cl_table, 1 << 7, 7, cl, 19, plain_entry)) return -1;
    for (int i = 0; i < hlit + hdist;) {
      refill_slow(z);
      uint32_t e = decode_sym(z, cl_table, 7);
      if (E_KIND(e) != K_LIT) return -1;
      int sym = (int)E_VAL(e), rep, v = 0;
      if (sym < 16) {
        lens[i++] = (uint8_t)sym;
        continue;
      }
      if (sym == 16) {
        if (i == 0) return -1;
        v = lens[i - 1];
        rep = 3 + (int)take_bits(z, 2);
      } else if (sym == 17) {
        rep = 3 + (int)take_bits(z, 3);
      } else {
        rep = 11 + (int)take_bits(z, 7);
      }
      if (i + rep > hlit + hdist) return -1;
      while (rep--) lens[i++] = (uint8_t)v;
    }
    if (overran(z)) return -1;
    if (lens[256] == 0) return -1; /* no end-of-block code */
    if (build_table(z->dyn_litlen, QKH_LITLEN_TABLE, LITLEN_BITS, lens, hlit, litlen_entry)) return -1;
    if (build_table(z->dyn_dist, QKH_DIST_TABLE, DIST_BITS, lens + hlit, hdist, dist_entry)) return -1;
    z->litlen = z->dyn_litlen;
    z->dist = z->dyn_dist;
    z->multi = NULL;
    z->dyn_multi_ready = 0;
    z->state = QKH_Z_CODES;
    return 0;
  }
  return -1;
}

/* ------------------------------------------------------------------- API */
void qkh_inflate_init(qkh_inflate *z, const uint8_t *data, size_t len) {
  memset(z, 0, sizeof *z);
  z->in = z->base = data;
  z->in_end = data + len;
  z->state = QKH_Z_MEMBER;
}

void qkh_inflate_init_at(qkh_inflate *z, const uint8_t *data, size_t len, uint64_t bit, size_t member_out,
                         unsigned members, int base_unknown) {
  qkh_inflate_init(z, data, len);
  z->in = data + (bit >> 3);
  if (bit & 7) {
    refill_slow(z);
    DROP(z, bit & 7);
  }
  z->state = QKH_Z_BLOCK;
  z->member_out = member_out;
  z->members = members;
  z->base_unknown = base_unknown;
}

/* Where could a block start?  Only dynamic-Huffman headers are looked for
 * (anything else carries too little structure to recognise; a slice that
 * really starts with a stored or fixed block is simply decoded in order by
 * its predecessor's thread, see pinflate.c).  Cheap tests first: block type,
 * HLIT/HDIST ranges, a complete code-length code; survivors get the full
 * header parse, which insists on complete literal/length and distance codes. */
int64_t qkh_inflate_find_block(const uint8_t *data, size_t len, uint64_t from_bit, uint64_t to_bit,
                               qkh_inflate *scratch) {
  static const uint8_t kraft[8] = {0, 64, 32, 16, 8, 4, 2, 1};   /* 2^(7-len) */
  const uint64_t last = len >= 24 ? (uint64_t)(len - 24) * 8u : 0;   /* keep the unaligned loads inside */
  if (to_bit > last) to_bit = last;
  for (uint64_t bit = from_bit; bit < to_bit; bit++) {
    const uint8_t *p = data + (bit >> 3);
    const unsigned sh = (unsigned)(bit & 7);
    const uint64_t w = load64(p) >> sh;              /* >= 57 bits */
    if ((w & 6u) != 4u) continue;                    /* BTYPE == 2 */
    if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
    const unsigned hclen = (unsigned)((w >> 13) & 15u) + 4u;
    /* code-length code lengths: hclen x 3 bits from bit 17 */
    uint64_t cl = (w >> 17) | ((load64(p + 7) >> sh) << 39);   /* bits 17.. : 39 from w, rest from p+7 */
    unsigned sum = 0, nz = 0;
    for (unsigned i = 0; i < hclen; i++) {
      const unsigned l = (unsigned)(cl & 7u);
      cl >>= 3;
      sum += kraft[l];
      nz += l != 0;
    }
    if (sum != 128u && !(nz == 1 && sum == 64u)) continue;
    qkh_inflate_init_at(scratch, data, len, bit, 0, 1, 1);
    if (qkh_read_block_header(scratch) == 0 && !overran(scratch)) return (int64_t)bit;
  }
  return -1;
}

/* the literal-run table of the current literal/length code (qkh_inflate::multi): for every literal code, then every pair and
 * triple of them that fits MULTI_BITS bits, one strided fill — an index belongs to the longest run it starts with, so the
 * shorter runs are written first and overwritten (a lookup chain per index took 17 % of a FASTQ's decode time: blocks are
 * ~30 KB) */
static void build_multi(uint32_t *multi, const uint32_t *lt) {
  struct lit_code { uint16_t code; uint8_t len, val; } c[256];
  int n = 0;
  memset(multi, 0, sizeof(uint32_t) << MULTI_BITS);
  for (uint32_t idx = 0; idx < (1u << LITLEN_BITS); idx++) {
    const uint32_t e = lt[idx];
    if (E_KIND(e) == K_LIT && E_LEN(e) != 0 && idx < (1u << E_LEN(e)) && n < 256) {
      c[n].code = (uint16_t)idx;     /* (the entry of a code is replicated at idx + k * 2^len: the lowest one names it) */
      c[n].len = (uint8_t)E_LEN(e);
      c[n].val = (uint8_t)E_VAL(e);
      n++;
    }
  }
  {   /* by code length (counting sort: lengths 1..11): the loops below stop at the first code that does not fit */
    int at[LITLEN_BITS + 2] = {0};
    struct lit_code t[256];
    for (int i = 0; i < n; i++) at[c[i].len + 1]++;
    for (int l = 1; l <= LITLEN_BITS + 1; l++) at[l] += at[l - 1];
    for (int i = 0; i < n; i++) t[at[c[i].len]++] = c[i];
    memcpy(c, t, (size_t)n * sizeof c[0]);
  }
#define FILL(bits, used, entry)                                                         \
  for (uint32_t i_ = (bits); i_ < (1u << MULTI_BITS); i_ += 1u << (used)) multi[i_] = (entry)
  for (int a = 0; a < n; a++) {
    const uint32_t u1 = c[a].len, b1 = c[a].code, v1 = c[a].val;
    FILL(b1, u1, u1 | (1u << 6) | (v1 << 8));
    for (int b = 0; b < n; b++) {
      const uint32_t u2 = u1 + c[b].len;
      if (u2 > MULTI_BITS) break;
      const uint32_t b2 = b1 | ((uint32_t)c[b].code << u1), v2 = v1 | ((uint32_t)c[b].val << 8);
      FILL(b2, u2, u2 | (2u << 6) | (v2 << 8));
      if (u2 + 1 > MULTI_BITS) continue;
      for (int d = 0; d < n; d++) {
        const uint32_t u3 = u2 + c[d].len;
        if (u3 > MULTI_BITS) break;
        FILL(b2 | ((uint32_t)c[d].code << u2), u3, u3 | (3u << 6) | ((v2 | ((uint32_t)c[d].val << 16)) << 8));
      }
    }
  }
#undef FILL
}
static void ensure_multi(qkh_inflate *z) {
  if (z->litlen == z->fixed_litlen) {
    if (!z->fixed_multi_ready) {
      build_multi(z->fixed_multi, z->fixed_litlen);
      z->fixed_multi_ready = 1;
    }
    z->multi = z->fixed_multi;
  } else {
    if (!z->dyn_multi_ready) {
      build_multi(z->dyn_multi, z->dyn_litlen);
      z->dyn_multi_ready = 1;
    }
    z->multi = z->dyn_multi;
  }
}

uint64_t qkh_inflate_bitpos(const qkh_inflate *z) {
  return (uint64_t)(z->in - z->base) * 8u - (uint64_t)z->bitcnt;
}

#define OUT_T uint8_t
#define READ_FN qkh_inflate_read
#define COPY_FN copy_match8
#include "inflate_body.inc"
#undef OUT_T
#undef READ_FN
#undef COPY_FN

#define OUT_T uint16_t
#define READ_FN qkh_inflate_read16
#define COPY_FN copy_match16
#include "inflate_body.inc"
#undef OUT_T
#undef READ_FN
#undef COPY_FN

void qkh_inflate_clone(qkh_inflate *dst, const qkh_inflate *src) {
  *dst = *src;
  if (src->litlen == src->fixed_litlen) dst->litlen = dst->fixed_litlen;
  else if (src->litlen == src->dyn_litlen) dst->litlen = dst->dyn_litlen;
  if (src->dist == src->fixed_dist) dst->dist = dst->fixed_dist;
  else if (src->dist == src->dyn_dist) dst->dist = dst->dyn_dist;
  if (src->multi == src->fixed_multi) dst->multi = dst->fixed_multi;
  else if (src->multi == src->dyn_multi) dst->multi = dst->dyn_multi;
}

/* ------------------------------------------------------------ member ends */
int qkh_end_list_take(qkh_end_list *l, const qkh_inflate *z, size_t before) {
  for (unsigned i = 0; i < z->tl_n; i++) {
    if (l->n == l->cap) {
      const unsigned cap = l->cap ? l->cap * 2 : 8;
      qkh_member_end *ne = realloc(l->ends, cap * sizeof *ne);
      if (ne) l->ends = ne;
      uint32_t *nc = realloc(l->piece_crc, ((size_t)cap + 1) * sizeof *nc);
      if (nc) l->piece_crc = nc;
      if (!ne || !nc) return -1;
      l->cap = cap;
    }
    l->ends[l->n] = z->tl[i];
    l->ends[l->n++].off += before;
  }
  return 0;
}

int qkh_end_list_crcs(qkh_end_list *l, const uint8_t *data, size_t len) {
  size_t from = 0;
  if (!l->piece_crc && !(l->piece_crc = malloc(sizeof *l->piece_crc))) return -1;
  for (unsigned i = 0; i <= l->n; i++) {
    size_t to = i < l->n ? l->ends[i].off : len;
    if (to > len) to = len;
    const uint32_t c = qkh_crc32(0u, data + from, to - from);
    l->piece_crc[i] = c;
    from = to;
  }
  return 0;
}

void qkh_end_list_free(qkh_end_list *l) {
  free(l->ends);
  free(l->piece_crc);
  memset(l, 0, sizeof *l);
}

/* ------------------------------------------------------------- line index */
size_t qkh_index_lines(const uint8_t *data, size_t len, uint32_t **nl, size_t *cap) {
  size_t n = 0;
  const uint8_t *p = data, *end = data + len;
  if (len >= 0xFFFFFFFFull) return (size_t)-1;
  while (p < end) {
    const uint8_t *q = memchr(p, '\n', (size_t)(end - p));
    if (!q) break;
    if (n == *cap) {
      /* FASTQ: a line per ~75 bytes; grow generously, rarely */
      size_t c = *cap ? *cap * 2 : (len / 48 + 1024);
      uint32_t *g = realloc(*nl, c * sizeof *g);
      if (!g) return (size_t)-1;
      *nl = g;
      *cap = c;
    }
    (*nl)[n++] = (uint32_t)(q - data);
    p = q + 1;
  }
  return n;
}
