/*
 * quack_host.h — host side of the drop-in (plain C, no GPU code).
 *
 * Everything quack does around the accumulation loop, re-stated:
 *   tokenizer   klib kseq_read as instantiated at quack.c:152 (gzread feed)
 *   adapters    read_adapters            quack.c:154-178  -> 2^20-bit bitset
 *   transform   transform()              quack.c:230-293
 *   draw / svg  draw(), svg.c            quack.c:295-856, svg.c:12-104
 *   CLI         parse_options(), main()  quack.c:59-132, 858-928
 * The accumulation itself is only ever done through include/quack_hip.h.
 */
#ifndef QUACK_HOST_H
#define QUACK_HOST_H

#include <stdint.h>
#include <stdio.h>

#include "quack_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tokenizer / batcher ------------------------------------------------ */
typedef struct qkh_reader qkh_reader;

qkh_reader *qkh_reader_open(const char *path);
void qkh_reader_close(qkh_reader *r);

/* Append whole records to a batch in the C-ABI layout until the stream ends
 * or the batch is full.  offsets[0] must be writable; on return
 * offsets[0..n] are set and *total_bytes == offsets[n].  *uniform_len is the
 * common read length if all n reads share one (so the caller may commit a
 * fixed-length batch), else 0.  Returns n >= 0, or < 0 on error.  The stream
 * stops for good at the first malformed record, like the reference's
 * `while ((l = kseq_read(seq)) >= 0)` (quack.c:193). */
int64_t qkh_reader_fill(qkh_reader *r, uint8_t *seq, uint8_t *qual,
                        uint64_t *offsets, uint64_t cap_bytes,
                        uint64_t cap_reads, uint64_t *total_bytes,
                        uint32_t *uniform_len);
/* Same, into a gapped batch (qk_accum_commit_gapped): read i starts at
 * starts[i], a multiple of `align` (a power of two; 128 = one HBM cache line,
 * QK_BATCH_ALIGNED128), and is lengths[i] long; *extent_bytes is the end of the
 * last read.  The bytes between reads are left as they are. */
int64_t qkh_reader_fill_gapped(qkh_reader *r, uint8_t *seq, uint8_t *qual,
                               uint64_t *starts, uint32_t *lengths, uint64_t cap_bytes,
                               uint64_t cap_reads, uint64_t align,
                               uint64_t *extent_bytes, uint32_t *uniform_len);
/* Same, into a strided batch (qk_accum_commit_strided): read i is written at
 * i * stride and is lengths[i] long.  The batch ends in front of the first read
 * longer than the stride (it waits for the next batch: qkh_reader_parked_len);
 * returns 0 without consuming anything when that read is the first one.  The
 * bytes behind a read's last base, up to the stride, are set to 0xFF in both
 * arrays (QK_BATCH_NEUTRAL_PADS). */
int64_t qkh_reader_fill_strided(qkh_reader *r, uint8_t *seq, uint8_t *qual, uint32_t *lengths,
                                uint64_t cap_bytes, uint64_t cap_reads, uint32_t stride,
                                uint32_t *uniform_len);
/* length of the record that did not fit the last batch and opens the next one (0: none) */
uint64_t qkh_reader_parked_len(const qkh_reader *r);
/* 1 once the stream is exhausted (or stopped by a malformed record) */
int qkh_reader_done(const qkh_reader *r);
/* 1 when the stream ended because a decoder thread ran out of memory, not because the file did:
 * the counts so far are those of a truncated stream and must not be reported */
int qkh_reader_failed(const qkh_reader *r);

/* ---- adapters ------------------------------------------------------------ */
/* bitset: QK_KMER_TABLE_WORDS words, bit i <=> kmers[i] = 1 (quack.c:171) */
void qkh_adapter_insert(uint32_t *bitset, const uint8_t *seq, uint64_t len);
int qkh_read_adapters(const char *path, uint32_t *bitset);

/* ---- whole-file accumulation through the C-ABI --------------------------- */
/* devices: list of HIP device ids (n >= 1); batches are dealt round-robin and
 * the per-device tables are summed with one RCCL all-reduce.  On success
 * *bases_out is malloc'd (max_len entries; NULL when max_len == 0). */
/* CLI only: the process exits (by _exit) right after the results are printed, so a successful
 * qkh_accumulate_file leaves its accumulators, reader threads and the HIP runtime to the OS. */
void qkh_process_exits_after_this(int on);
int qkh_process_exits(void);   /* ... as the caller said */
int qkh_accumulate_file(const char *path, const uint32_t *bitset,
                        const int *devices, int n_devices,
                        qk_base_info **bases_out, uint64_t *max_len,
                        uint64_t *n_reads);
const char *qkh_last_error(void);
/* device ids from QUACK_DEVICES (comma separated; default: 0); returns how many */
int qkh_device_list(int *devs, int cap);

/* ---- post-processing and SVG --------------------------------------------- */
typedef struct {
  FILE *out;
  int depth;      /* svg.c:9 _svg_indent_level */
  int has_name;
} qkh_svg;

void qkh_transform(qk_base_info *bases, uint64_t *max_length_io,
                   uint64_t *original_max_length, uint64_t number_of_sequences,
                   FILE *err);
void qkh_draw(qkh_svg *w, const qk_base_info *bases, uint64_t max_length,
              uint64_t number_of_sequences, int position, int adapters_used);
void qkh_svg_begin(qkh_svg *w, FILE *out, int paired, int adapters, const char *name);
void qkh_svg_end(qkh_svg *w);

/* Render one complete document from raw counter tables (test entry point:
 * lets tests drive transform+draw with tables from any source).  rev may be
 * NULL (unpaired).  Tables are modified in place, like transform(). */
int qkh_render_document(FILE *out, FILE *err, const char *name, int adapters,
                        qk_base_info *fwd, uint64_t fwd_max_len, uint64_t fwd_reads,
                        qk_base_info *rev, uint64_t rev_max_len, uint64_t rev_reads);

/* The CLI (quack.c:858-928) as a function; returns the process exit code. */
int qkh_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif
