/* pinflate.h — multi-threaded decoder for one ordinary gzip stream (see pinflate.c). */
#ifndef QKH_PINFLATE_H
#define QKH_PINFLATE_H

#include <stddef.h>
#include <stdint.h>

#include "inflate_fast.h"

typedef struct qkh_pinflate qkh_pinflate;

/* `data` (the whole compressed file) must stay mapped until close. */
qkh_pinflate *qkh_pinflate_open(const uint8_t *data, size_t len, int threads, size_t slice_bytes);
/* Same contract as qkh_source_next: the decompressed stream, in order, one slice per call. */
int qkh_pinflate_next(qkh_pinflate *p, const uint8_t **data, size_t *len);
/* The gzip member trailers inside the slice handed out by the last qkh_pinflate_next (offsets into
 * it, stored CRC-32 / length verdict), and the CRC-32 of the n_ends + 1 pieces they cut it into:
 * the caller chains them across slices (crc32_combine) and compares at every member end. */
void qkh_pinflate_ends(qkh_pinflate *p, const qkh_member_end **ends, unsigned *n_ends, const uint32_t **piece_crc);
/* newline offsets of that slice (n == (size_t)-1: none) */
void qkh_pinflate_lines(qkh_pinflate *p, const uint32_t **nl, size_t *n);
/* After qkh_pinflate_next returned 0: 1 if the stream is NOT over but a slice outgrew the in-order
 * decoder's memory bound (a hostile or extremely compressible file).  *z then decodes on from the
 * exact point with the window's last *window_len bytes (<= 32768) as history: the caller goes on
 * with the constant-memory serial producer. */
int qkh_pinflate_handoff(qkh_pinflate *p, qkh_inflate *z, uint8_t *window, size_t *window_len);
/* 1 when a worker could not allocate what the in-order decode of a slice needs: the stream then ends
 * early, and the caller must report that instead of taking it for the end of the file */
int qkh_pinflate_failed(qkh_pinflate *p);
/* slices whose speculative decode was kept / that were decoded again in order */
void qkh_pinflate_stats(const qkh_pinflate *p, unsigned *kept, unsigned *redone);
void qkh_pinflate_close(qkh_pinflate *p);

#endif
