/* inflate_fast.h — gzip/DEFLATE decoder of the host feed (see inflate_fast.c). */
#ifndef QKH_INFLATE_FAST_H
#define QKH_INFLATE_FAST_H

#include <stddef.h>
#include <stdint.h>

enum { QKH_LITLEN_TABLE = 2048 + 4096, QKH_DIST_TABLE = 256 + 2048 };
enum { QKH_Z_MEMBER, QKH_Z_BLOCK, QKH_Z_STORED, QKH_Z_CODES, QKH_Z_TRAILER, QKH_Z_DONE, QKH_Z_ERROR };

typedef struct {
  const uint8_t *in, *in_end;   /* the whole compressed file (memory mapped) */
  uint64_t bitbuf;
  int bitcnt;
  int state, final_block, fixed_ready;
  uint32_t stored_left;
  uint32_t pend_len, pend_dist; /* match cut by the end of an output block */
  size_t member_out;            /* bytes produced by the current member (ISIZE check) */
  unsigned members;
  const uint32_t *litlen, *dist;
  uint32_t fixed_litlen[QKH_LITLEN_TABLE], fixed_dist[QKH_DIST_TABLE];
  uint32_t dyn_litlen[QKH_LITLEN_TABLE], dyn_dist[QKH_DIST_TABLE];
} qkh_inflate;

void qkh_inflate_init(qkh_inflate *z, const uint8_t *data, size_t len);

/* Produce up to `cap` bytes at `out`.  The `history` bytes before `out` must be
 * the previously produced output (up to 32768 are ever referenced).  Returns
 * the number of bytes written; 0 when the stream is finished; -1 on a format
 * error with nothing produced (bytes produced before an error are returned
 * first, like gzread). */
long qkh_inflate_read(qkh_inflate *z, uint8_t *out, size_t cap, size_t history);

#endif
