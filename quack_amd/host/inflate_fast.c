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
 * The trailer's ISIZE is checked; the CRC-32 is not (the reference does not
 * surface zlib's CRC errors either: a failing gzread just ends its read loop,
 * quack.c:193).  Anything this decoder rejects makes the caller stop exactly
 * there, like a failing gzread.  tests/test_inflate.py fuzzes it against zlib.
 */
#include "inflate_fast.h"

#include <string.h>

#define LITLEN_BITS 11
#define DIST_BITS 8
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

static int read_block_header(qkh_inflate *z) {
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
    z->state = QKH_Z_CODES;
    return 0;
  }
  return -1;
}

/* ------------------------------------------------------------------- API */
void qkh_inflate_init(qkh_inflate *z, const uint8_t *data, size_t len) {
  memset(z, 0, sizeof *z);
  z->in = data;
  z->in_end = data + len;
  z->state = QKH_Z_MEMBER;
}

/* copy `len` (3..258) bytes from `dist` back; regions may overlap (dist < len).
 * May write up to 15 bytes past dst+len (the fast loop keeps that slack). */
static inline void copy_match(uint8_t *dst, size_t dist, size_t len) {
  const uint8_t *src = dst - dist;
  if (dist >= 8) {
    /* most FASTQ matches are short: two unconditional 8-byte moves cover <= 16 */
    memcpy(dst, src, 8);
    memcpy(dst + 8, src + 8, 8);
    if (len > 16) {
      uint8_t *end = dst + len;
      dst += 16;
      src += 16;
      do {
        memcpy(dst, src, 8);
        dst += 8;
        src += 8;
      } while (dst < end);
    }
  } else if (dist == 1) {
    memset(dst, *src, len);   /* runs (quality plateaus) */
  } else {
    for (size_t i = 0; i < len; i++) dst[i] = src[i];
  }
}

long qkh_inflate_read(qkh_inflate *z, uint8_t *out, size_t cap, size_t history) {
  uint8_t *o = out, *const o_end = out + cap;
  uint8_t *mark = out;   /* output already added to member_out */
#define ACCOUNT() (z->member_out += (size_t)(o - mark), mark = o)
  if (z->state == QKH_Z_DONE || z->state == QKH_Z_ERROR) return z->state == QKH_Z_DONE ? 0 : -1;
  for (;;) {
    switch (z->state) {
      case QKH_Z_MEMBER:
        if (z->in >= z->in_end) {
          z->state = QKH_Z_DONE;
          return (long)(o - out);
        }
        if (read_gzip_header(z)) {
          /* trailing garbage after at least one member ends the stream (zlib
           * behaves the same); garbage instead of a first member is an error */
          z->state = z->members ? QKH_Z_DONE : QKH_Z_ERROR;
          return (z->state == QKH_Z_DONE || o > out) ? (long)(o - out) : -1;
        }
        z->state = QKH_Z_BLOCK;
        break;
      case QKH_Z_BLOCK:
        if (read_block_header(z) || overran(z)) goto fail;
        break;
      case QKH_Z_STORED: {
        size_t n = z->stored_left, room = (size_t)(o_end - o), avail = (size_t)(z->in_end - z->in);
        if (n > room) n = room;
        if (n > avail) goto fail;
        memcpy(o, z->in, n);
        o += n;
        z->in += n;
        z->stored_left -= (uint32_t)n;
        if (z->stored_left) goto out_full;
        z->state = z->final_block ? QKH_Z_TRAILER : QKH_Z_BLOCK;
        break;
      }
      case QKH_Z_CODES: {
        const uint32_t *lt = z->litlen, *dt = z->dist;
        /* a match cut by the end of the previous output block */
        if (z->pend_len) {
          size_t n = z->pend_len, room = (size_t)(o_end - o);
          if (n > room) n = room;
          for (size_t i = 0; i < n; i++) o[i] = o[(ptrdiff_t)i - (ptrdiff_t)z->pend_dist];
          o += n;
          z->pend_len -= (uint32_t)n;
          if (z->pend_len) goto out_full;
        }
        for (;;) {
          /* fast loop: >= 8 input bytes for the refill, room for a longest
           * match plus the copy's overshoot */
          while (z->in_end - z->in >= NEED_INPUT_SLACK && o_end - o >= 258 + 32) {
            uint32_t e;
            z->bitbuf |= load64(z->in) << z->bitcnt;
            z->in += (63 - z->bitcnt) >> 3;
            z->bitcnt |= 56;
            e = lt[PEEK(z, LITLEN_BITS)];
            if (__builtin_expect(E_KIND(e) == K_LIT, 1)) {
              /* literal run: primary-table literals take <= 11 bits each, so
               * five of them fit in the >= 56 bits of one refill */
              DROP(z, E_LEN(e));
              *o++ = (uint8_t)E_VAL(e);
              e = lt[PEEK(z, LITLEN_BITS)];
              if (E_KIND(e) == K_LIT) {
                DROP(z, E_LEN(e));
                *o++ = (uint8_t)E_VAL(e);
                e = lt[PEEK(z, LITLEN_BITS)];
                if (E_KIND(e) == K_LIT) {
                  DROP(z, E_LEN(e));
                  *o++ = (uint8_t)E_VAL(e);
                  e = lt[PEEK(z, LITLEN_BITS)];
                  if (E_KIND(e) == K_LIT) {
                    DROP(z, E_LEN(e));
                    *o++ = (uint8_t)E_VAL(e);
                    e = lt[PEEK(z, LITLEN_BITS)];
                    if (E_KIND(e) == K_LIT) {
                      DROP(z, E_LEN(e));
                      *o++ = (uint8_t)E_VAL(e);
                    }
                  }
                }
              }
              continue;
            }
            if (E_KIND(e) == K_SUB) {
              DROP(z, LITLEN_BITS);
              e = lt[E_VAL(e) + PEEK(z, E_EXTRA(e))];
              if (E_KIND(e) == K_LIT) {
                DROP(z, E_LEN(e));
                *o++ = (uint8_t)E_VAL(e);
                continue;
              }
            }
            DROP(z, E_LEN(e));
            if (E_KIND(e) == K_BASE) {
              size_t len = E_VAL(e) + PEEK(z, E_EXTRA(e)), dist;
              uint32_t d;
              DROP(z, E_EXTRA(e));
              d = decode_sym(z, dt, DIST_BITS);
              if (E_KIND(d) != K_BASE) goto fail;
              dist = E_VAL(d) + PEEK(z, E_EXTRA(d));
              DROP(z, E_EXTRA(d));
              if (dist > (size_t)(o - out) + history) goto fail;
              copy_match(o, dist, len);
              o += len;
              continue;
            }
            if (E_KIND(e) == K_END) goto block_done;
            goto fail;
          }
          /* careful step: one symbol with every bound checked */
          {
            uint32_t e;
            if (o == o_end) goto out_full;
            refill_slow(z);
            e = decode_sym(z, lt, LITLEN_BITS);
            if (overran(z)) goto fail;
            if (E_KIND(e) == K_LIT) {
              *o++ = (uint8_t)E_VAL(e);
            } else if (E_KIND(e) == K_BASE) {
              size_t len = E_VAL(e) + PEEK(z, E_EXTRA(e)), dist, n, room;
              uint32_t d;
              DROP(z, E_EXTRA(e));
              refill_slow(z);
              d = decode_sym(z, dt, DIST_BITS);
              if (E_KIND(d) != K_BASE) goto fail;
              dist = E_VAL(d) + PEEK(z, E_EXTRA(d));
              DROP(z, E_EXTRA(d));
              if (overran(z)) goto fail;
              if (dist > (size_t)(o - out) + history) goto fail;
              room = (size_t)(o_end - o);
              n = len < room ? len : room;
              for (size_t i = 0; i < n; i++) o[i] = o[(ptrdiff_t)i - (ptrdiff_t)dist];
              o += n;
              if (n < len) {
                z->pend_len = (uint32_t)(len - n);
                z->pend_dist = (uint32_t)dist;
                goto out_full;
              }
            } else if (E_KIND(e) == K_END) {
              goto block_done;
            } else {
              goto fail;
            }
          }
        }
      block_done:
        z->state = z->final_block ? QKH_Z_TRAILER : QKH_Z_BLOCK;
        break;
      }
      case QKH_Z_TRAILER: {
        ACCOUNT();
        align_to_byte(z);
        if (z->in > z->in_end || z->in_end - z->in < 8) goto fail;
        uint32_t isize = z->in[4] | ((uint32_t)z->in[5] << 8) | ((uint32_t)z->in[6] << 16) | ((uint32_t)z->in[7] << 24);
        if (isize != (uint32_t)z->member_out) goto fail;
        z->in += 8;
        z->members++;
        z->state = QKH_Z_MEMBER;
        break;
      }
      default:
        goto fail;
    }
  }
out_full:
  ACCOUNT();
  return (long)(o - out);
fail:
  ACCOUNT();
  z->state = QKH_Z_ERROR;
  /* what was produced before the error is still delivered, like gzread */
  return o > out ? (long)(o - out) : -1;
#undef ACCOUNT
}
