/* pinflate.h — multi-threaded decoder for one ordinary gzip stream (see pinflate.c). */
#ifndef QKH_PINFLATE_H
#define QKH_PINFLATE_H

#include <stddef.h>
#include <stdint.h>

typedef struct qkh_pinflate qkh_pinflate;

/* `data` (the whole compressed file) must stay mapped until close. */
qkh_pinflate *qkh_pinflate_open(const uint8_t *data, size_t len, int threads, size_t slice_bytes);
/* Same contract as qkh_source_next: the decompressed stream, in order, one slice per call. */
int qkh_pinflate_next(qkh_pinflate *p, const uint8_t **data, size_t *len);
/* slices whose speculative decode was kept / that were decoded again in order */
void qkh_pinflate_stats(const qkh_pinflate *p, unsigned *kept, unsigned *redone);
void qkh_pinflate_close(qkh_pinflate *p);

#endif
