/* source.h — decompressed-byte source of the tokenizer (see source.c). */
#ifndef QKH_SOURCE_H
#define QKH_SOURCE_H

#include <stddef.h>
#include <stdint.h>

typedef struct qkh_source qkh_source;

qkh_source *qkh_source_open(const char *path);
/* Release the block handed out by the previous call (if any) and wait for the
 * next one, in stream order.  Returns 1 with *data / *len set, or 0 at the end
 * of the stream (EOF or the first undecodable byte, like a failing gzread). */
int qkh_source_next(qkh_source *s, const uint8_t **data, size_t *len);
/* Offsets of the newlines inside the block handed out by the last qkh_source_next, ascending, relative to
 * its data pointer — indexed by the producer threads.  Returns 0 when there is no index for this block
 * (zlib / plain producers, the last few bytes of a stream): the caller looks for the lines itself. */
int qkh_source_lines(qkh_source *s, const uint32_t **nl, size_t *n);
/* 1 when a producer thread ran out of memory: the stream ended early for that reason, not because the
 * file did — the caller must fail instead of reporting a truncated stream's counts. */
int qkh_source_failed(const qkh_source *s);
void qkh_source_close(qkh_source *s);
/* "zlib", "inflate_fast", "bgzf xN" or "plain": which producer is running */
const char *qkh_source_kind(const qkh_source *s);

#endif
