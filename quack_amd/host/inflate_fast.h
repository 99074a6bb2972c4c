/* inflate_fast.h — gzip/DEFLATE decoder of the host feed (see inflate_fast.c). */
#ifndef QKH_INFLATE_FAST_H
#define QKH_INFLATE_FAST_H

#include <stddef.h>
#include <stdint.h>

enum { QKH_LITLEN_TABLE = 2048 + 4096, QKH_DIST_TABLE = 256 + 2048 };
#ifndef QKH_MULTI_BITS
#define QKH_MULTI_BITS 12   /* index width of the literal-run table (qkh_inflate::multi); 13 measured slower on both hosts */
#endif
enum { QKH_Z_MEMBER, QKH_Z_BLOCK, QKH_Z_STORED, QKH_Z_CODES, QKH_Z_TRAILER, QKH_Z_DONE, QKH_Z_ERROR };

/* A member trailer met while decoding: `off` output elements into the CALL that met it
 * (the caller adds what it had before), the CRC-32 stored in the file, and whether the
 * stored length disagreed with the bytes produced.  The decoder checks neither against
 * the data: the bytes of a member may come from several decoders (pinflate.c), so the
 * caller chains the CRCs of the pieces (source.c) — zlib's gzread, which the reference
 * reads through, fails at such a trailer (quack.c:187,193). */
typedef struct {
  size_t off;
  uint32_t crc;
  uint32_t bad_length;
} qkh_member_end;
enum { QKH_TRAILER_LOG = 64 };

typedef struct {
  const uint8_t *in, *in_end;   /* the whole compressed file (memory mapped) */
  uint64_t bitbuf;
  int bitcnt;
  int state, final_block, fixed_ready;
  uint32_t stored_left;
  uint32_t pend_len, pend_dist; /* match cut by the end of an output block */
  size_t member_out;            /* bytes produced by the current member (ISIZE check) */
  unsigned members;
  /* decoding a slice of the stream (pinflate.c) */
  const uint8_t *base;          /* start of the compressed data: bit positions count from here */
  uint64_t stop_bit;            /* != 0: stop in front of the first block header at or past this bit */
  int stopped;                  /* ... which has happened */
  int base_unknown;             /* started inside a member of unknown length so far */
  int pend_set;                 /* the first trailer met in that state: its ISIZE and the */
  uint32_t pend_isize;          /*   bytes this decoder had produced for the member by then */
  size_t pend_out;
  /* trailers met by the current call (the caller clears tl_n before a call; a full log ends the call) */
  unsigned tl_n;
  qkh_member_end tl[QKH_TRAILER_LOG];
  const uint32_t *litlen, *dist;
  uint32_t fixed_litlen[QKH_LITLEN_TABLE], fixed_dist[QKH_DIST_TABLE];
  uint32_t dyn_litlen[QKH_LITLEN_TABLE], dyn_dist[QKH_DIST_TABLE];
  /* literal runs (round 4): what the next 12 bits hold when they START with literals — up to three of them per lookup
   * (FASTQ is nearly all literals: bases take 2-3 bits, scores 5-6).  bits 0-5 the bits to consume, 6-7 the count
   * (0: no literal there, take the ordinary table), 8-31 the literals.  Built from `litlen` when a block's first symbols
   * are decoded (not when its header is parsed: the search for block starts parses many headers it never decodes). */
  const uint32_t *multi;
  int fixed_multi_ready, dyn_multi_ready;
  uint32_t fixed_multi[1 << QKH_MULTI_BITS], dyn_multi[1 << QKH_MULTI_BITS];
} qkh_inflate;

/* The member ends of one delivered chunk of output (a ring block, a pinflate slice) and the
 * CRC-32 of the n + 1 pieces they cut the chunk into. */
typedef struct {
  qkh_member_end *ends;
  uint32_t *piece_crc;
  unsigned n, cap;
} qkh_end_list;
/* A decoder call ends when its trailer log is full.  If that call produced nothing — 64 members in a
 * row without a byte of output between them — it returns 0 without the stream being over: the caller
 * takes the log and calls again (zlib's gzread, which the reference reads through, walks through any
 * number of empty members). */
static inline int qkh_inflate_log_full(const qkh_inflate *z) {
  return z->tl_n == QKH_TRAILER_LOG && z->state != QKH_Z_DONE && z->state != QKH_Z_ERROR;
}
/* append the trailers the last decoder call logged; `before` = chunk bytes produced before that call */
int qkh_end_list_take(qkh_end_list *l, const qkh_inflate *z, size_t before);
/* fill piece_crc[0..n] from the chunk's final bytes */
int qkh_end_list_crcs(qkh_end_list *l, const uint8_t *data, size_t len);
void qkh_end_list_free(qkh_end_list *l);

/* zlib's crc32() with carry-less multiplies where the CPU has them (crc32_fold.c) */
uint32_t qkh_crc32(uint32_t crc, const uint8_t *buf, size_t len);

/* Offsets of every '\n' in data[0..len) (len < 4 GiB), in a buffer that grows as needed: the
 * producers index the lines of their chunks on their own threads, so that the tokenizer — one
 * thread, and the end-to-end limiter — does not have to look for them.  Returns the count, or
 * (size_t)-1 when there is no index (allocation failure, oversized chunk). */
size_t qkh_index_lines(const uint8_t *data, size_t len, uint32_t **nl, size_t *cap);

void qkh_inflate_init(qkh_inflate *z, const uint8_t *data, size_t len);
/* a copy that decodes on from where `src` stands (the table pointers are re-aimed at the copy) */
void qkh_inflate_clone(qkh_inflate *dst, const qkh_inflate *src);

/* Produce up to `cap` bytes at `out`.  The `history` bytes before `out` must be
 * the previously produced output (up to 32768 are ever referenced).  Returns
 * the number of bytes written; 0 when the stream is finished; -1 on a format
 * error with nothing produced (bytes produced before an error are returned
 * first, like gzread). */
long qkh_inflate_read(qkh_inflate *z, uint8_t *out, size_t cap, size_t history);

/* --- decoding from the middle of a stream (speculative parallel inflate) --- */
/* Start at bit `bit` of `data`, which must be the first bit of a block header
 * inside a member that has produced `member_out` bytes so far.  With
 * `base_unknown` the member's earlier length is not known: matches may reach
 * anywhere into the caller's history and the first ISIZE met is reported in
 * pend_* instead of checked. */
void qkh_inflate_init_at(qkh_inflate *z, const uint8_t *data, size_t len, uint64_t bit, size_t member_out,
                         unsigned members, int base_unknown);
/* Same decoder over 16-bit elements: values < 256 are bytes, anything else was
 * copied out of the caller's history (which the caller fills with markers). */
long qkh_inflate_read16(qkh_inflate *z, uint16_t *out, size_t cap, size_t history);
uint64_t qkh_inflate_bitpos(const qkh_inflate *z);
int qkh_read_block_header(qkh_inflate *z);
/* First bit position in [from_bit, to_bit) where a dynamic-Huffman block header
 * with complete codes parses, or -1.  `scratch` is clobbered. */
int64_t qkh_inflate_find_block(const uint8_t *data, size_t len, uint64_t from_bit, uint64_t to_bit,
                               qkh_inflate *scratch);

#endif
