/*
 * quack_hip.h — C-ABI of the MI355X accumulation path ("libquack_hip.so").
 *
 * This is the drop-in boundary for quack's per-read statistics loop.  The
 * reference has no plugin API; the seam is two C functions and two structs:
 *
 *     int*           read_adapters(char *adapters_file)          quack.c:154-178
 *     sequence_data* read_fastq(char *fastq_file, int *kmers)    quack.c:180-228
 *     base_information / sequence_data                           quack.c:134-146
 *
 * Everything below is `extern "C"`, plain pointers and sizes, no C++ or torch
 * types.  A host (the C CLI in quack_amd/host, the Python mirror in
 * quack_amd/, or quack.c itself via the stub shown in INTEGRATION.md) tokenises
 * FASTQ on CPU cores and hands *read batches* to an accumulator; the
 * accumulator owns device memory, pinned staging buffers and HIP streams, and
 * returns exactly the table read_fastq() would have produced.
 *
 * Batch layout (host and device, identical):
 *     seq [total]   sequence bytes of all reads, concatenated, no separators
 *     qual[total]   quality bytes, same offsets
 *     offsets[n+1]  u64 start offset of each read (offsets[n] == total);
 *                   NULL for a fixed-length batch (read r starts at r*read_len)
 * (a third, gapped form — starts[n] + lengths[n], reads padded onto 128-byte
 * cache lines — is described with qk_accum_submit_device_gapped below)
 * Algorithmic HBM traffic: 2 bytes per base (+ 8 bytes per read when ragged,
 * + 12 when gapped).
 *
 * Result layout: `qk_base_info`, bit-for-bit the reference's base_information
 * (97 x u64 = 776 bytes per position, quack.c:134-139).
 *
 * Error convention: every function returns 0 on success and a negative
 * QK_E* code on failure; qk_last_error() gives the message (thread-local).
 * There is NO CPU fallback: without a usable HIP device every accumulator
 * call fails with QK_ENODEV.
 */
#ifndef QUACK_HIP_H
#define QUACK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QK_N_SCORES 91            /* quack.c:135  scores[91]            */
#define QK_N_BASES 4              /* quack.c:136  content[4]: A,T,C,G   */
#define QK_ROW_LENGTH 95          /* quack.c:137  length_count          */
#define QK_ROW_KMER 96            /* quack.c:138  kmer_count            */
#define QK_N_ROWS 97              /* u64 words per position             */
#define QK_KMER_SIZE 10           /* quack.c:155,184                    */
#define QK_KMER_TABLE_BITS (1u << 20)              /* 4^10 (quack.c:156) */
#define QK_KMER_TABLE_WORDS (QK_KMER_TABLE_BITS / 32) /* 128 KiB bitset  */
#define QK_TAIL_SLACK 16          /* readable bytes required after seq/qual */

#define QK_OK 0
#define QK_EINVAL (-1)            /* bad argument                        */
#define QK_ENODEV (-2)            /* no HIP device / extension unusable  */
#define QK_EHIP (-3)              /* a HIP runtime call failed           */
#define QK_ENOMEM (-4)
#define QK_ERCCL (-5)             /* an RCCL call failed                 */
#define QK_ESTATE (-6)            /* call sequence violated              */

/* Same memory image as the reference's base_information (quack.c:134-139). */
typedef struct {
  uint64_t scores[QK_N_SCORES];
  uint64_t content[QK_N_BASES];
  uint64_t length_count;
  uint64_t kmer_count;
} qk_base_info;

/* Opaque accumulator: the state read_fastq() keeps in `bases`, `max_length`
 * and `number_of_sequences` (quack.c:186-191), resident on one GPU. */
typedef struct qk_accum qk_accum;

/* ---- library ---------------------------------------------------------- */
const char *qk_last_error(void);
const char *qk_version(void);
int qk_device_count(int *count);

/* ---- accumulator life cycle  (replaces read_fastq, quack.c:180-228) ---- */

/* kmer_bitset: QK_KMER_TABLE_WORDS words, bit i set <=> kmers[i] != 0 in the
 * reference's table (quack.c:162-171), or NULL when no -a was given
 * (kmers == NULL, quack.c:210).  max_len_hint sizes the first table
 * allocation; the table grows when a longer read arrives (quack.c:194-198). */
int qk_accum_create(qk_accum **out, int device, const uint32_t *kmer_bitset,
                    uint64_t max_len_hint);
void qk_accum_destroy(qk_accum *acc);

/* Zero-copy feed: borrow a pinned host batch buffer (double-buffered; blocks
 * until one is free), fill it, commit it.  Capacities are in bytes / reads. */
int qk_accum_acquire(qk_accum *acc, uint8_t **seq, uint8_t **qual,
                     uint64_t **offsets, uint64_t *cap_bytes,
                     uint64_t *cap_reads);
/* Slot size: 32 MiB per array by default (QUACK_HIP_BATCH_MB); a batch holds whole reads, so a
 * read longer than a slot needs bigger ones: drops both slots (waiting for their work) and makes
 * the next acquire allocate at least min_bytes.  No slot may be held.  Never shrinks. */
int qk_accum_resize_slots(qk_accum *acc, uint64_t min_bytes);
/* offsets_used == 0: fixed-length batch of n_reads x read_len. */
int qk_accum_commit(qk_accum *acc, uint64_t n_reads, uint64_t total_bytes,
                    int offsets_used, uint32_t read_len);

/* Copying feed from caller-owned host memory (any size; split internally). */
int qk_accum_submit(qk_accum *acc, const uint8_t *seq, const uint8_t *qual,
                    const uint64_t *offsets, uint64_t n_reads);
int qk_accum_submit_fixed(qk_accum *acc, const uint8_t *seq,
                          const uint8_t *qual, uint32_t read_len,
                          uint64_t n_reads);

/* Device-resident feed: pointers are device addresses (hipMalloc / a torch
 * CUDA tensor's data_ptr) with QK_TAIL_SLACK readable bytes after `total`.
 * d_offsets == NULL: fixed-length.  max_len: longest read in the batch.
 * hip_stream: a hipStream_t (NULL = the accumulator's own stream). The call
 * only enqueues; it does not synchronise. */
int qk_accum_submit_device(qk_accum *acc, const void *d_seq,
                           const void *d_qual, const void *d_offsets,
                           uint64_t n_reads, uint64_t total_bytes,
                           uint32_t max_len, void *hip_stream);

/* ---- gapped batches: reads need not be adjacent -----------------------------
 * Read r occupies bytes [starts[r], starts[r] + lengths[r]) of seq / qual;
 * starts ascend and reads do not overlap; `extent_bytes` is the end of the
 * last read.  Same counters as the packed form (nothing in the reference
 * corresponds: it never holds two reads at once, quack.c:193).  What it is for:
 * with QK_BATCH_ALIGNED128 the producer promises that every read starts on a
 * 128-byte boundary (one HBM cache line).  Long reads are cut into position
 * tiles that different workgroups take at different times; with aligned starts
 * a tile is whole cache lines and every line is fetched once (packed layout:
 * 5 lines where 4 would do).  The host tokenizer pads long-read batches this
 * way (quack_amd/host/pipeline.c).  A batch that breaks the promise is
 * detected: the next qk_accum_sync / finish fails with QK_EINVAL. */
#define QK_BATCH_ALIGNED128 1u

int qk_accum_submit_device_gapped(qk_accum *acc, const void *d_seq, const void *d_qual,
                                  const void *d_starts /* u64[n_reads] */,
                                  const void *d_lengths /* u32[n_reads] */,
                                  uint64_t n_reads, uint64_t extent_bytes,
                                  uint32_t max_len, uint32_t flags, void *hip_stream);
/* Pinned-slot form: between qk_accum_acquire and the commit, `offsets[r]` of
 * the slot holds starts[r] and the array returned here the lengths. */
int qk_accum_slot_lengths(qk_accum *acc, uint32_t **lengths);
int qk_accum_commit_gapped(qk_accum *acc, uint64_t n_reads, uint64_t extent_bytes, uint32_t flags);

/* ---- strided batches: fixed stride, per-read lengths -------------------------
 * Read r occupies bytes [r * stride, r * stride + lengths[r]) of seq / qual;
 * stride is a multiple of 4 and >= every length; the bytes behind a read's
 * last base are never counted.  This is the form for short reads that are
 * *nearly* all one length (adapter/quality-trimmed Illumina runs): the
 * kernels keep the address arithmetic and the software-pipelined loop of a
 * fixed-length batch and mask the tails (a packed ragged batch of the same
 * reads runs ~20 % slower).  The host tokenizer switches to it by itself
 * (quack_amd/host/pipeline.c).  Algorithmic traffic: 2 B/base + 4 B/read.
 * Pinned-slot form: lengths go into the array of qk_accum_slot_lengths. */
int qk_accum_submit_device_strided(qk_accum *acc, const void *d_seq, const void *d_qual,
                                   const void *d_lengths /* u32[n_reads], or NULL: all max_len (padded fixed length) */, uint64_t n_reads,
                                   uint32_t stride, uint32_t max_len, void *hip_stream);
int qk_accum_commit_strided(qk_accum *acc, uint64_t n_reads, uint32_t stride);
/* When to prefer the strided layout over a packed ragged batch for short reads of mixed lengths (<= 352 bases): nearly always.  The
 * strided kernels take n_reads x stride positions whatever the lengths; the packed ones pay per read and per base.  10M reads of
 * U[30,150] bases (mean 60 % of the stride): strided 0.534 ms, packed 0.559; with the adapter table 0.596 against 0.914
 * (profiles/r05_ragged_probe.log).  The host feed switches at a mean of 50 % of the longest read, 35 % with adapters. */
/* QK_BATCH_NEUTRAL_PADS (round 4): the producer promises that the bytes behind every read's last base, up to the stride,
 * are 0xFF in both arrays.  Such a byte counts into a quality row the flush discards and matches none of T / C / G, so
 * the kernel runs without tail masks (a third of its instructions; 10M trimmed 150 bp reads 0.5300 -> 0.5178 ms); a
 * position's content[A] comes from the lengths the kernel counts anyway.  With the adapter table loaded such a batch (stride a
 * multiple of 4, 64 .. 352) takes the 16-positions-per-lane kernel, rows of several reads (round 5: 0.44 -> 0.59 of the HBM
 * peak on trimmed 150 bp reads).  The host tokenizer writes its strided batches that way (quack_amd/host/reader.c) and
 * qk_accum_submit_strided neutralises the pads on its way into the pinned slot.
 * FAILURE MODE OF A BROKEN PROMISE: a pad byte that is T / C / G is counted into t / c / g while `valid` comes from the lengths,
 * so content[A] = valid - t - c - g is silently WRONG (it underflows), rc 0.  What guards against it: the pinned-slot commit
 * checks every pad byte of every read (QK_EINVAL otherwise); a device-resident batch has its pads verified by a kernel for the
 * first two such batches of an accumulator — a violation fails the next qk_accum_sync / finish — and for every batch with
 * QUACK_HIP_CHECK_PADS=1 (tests, debugging a producer); beyond that it is taken at its word.  Same counters as without the
 * flag. */
#define QK_BATCH_NEUTRAL_PADS 2u
int qk_accum_submit_device_strided_flags(qk_accum *acc, const void *d_seq, const void *d_qual, const void *d_lengths,
                                         uint64_t n_reads, uint32_t stride, uint32_t max_len, uint32_t flags, void *hip_stream);
int qk_accum_commit_strided_flags(qk_accum *acc, uint64_t n_reads, uint32_t stride, uint32_t flags);
/* Padded fixed-length batches (round 4): d_lengths == NULL above means "every read is max_len long" —
 * a fixed-length batch whose reads lie `stride` (a multiple of 4, >= max_len) bytes apart.  Uniform reads
 * whose length is not a multiple of 4 (150, 250, 125, 50 bp) start on odd byte phases when packed; padded,
 * every 8-byte chunk starts on a dword and the batch runs the dword-aligned kernels (with the adapter scan:
 * 16 positions per lane, one 16-byte load per lane and array).  The pad bytes are never counted.  Nothing in
 * the reference corresponds (it holds one read at a time, quack.c:193); the counters are those of the packed
 * batch.  Algorithmic traffic stays 2 B/base; the padding is 1.3 % more bytes at 150 bp.
 *   qk_accum_padded_stride: the stride the library wants for uniform reads of read_len on this accumulator
 *     (0: keep them packed — multiples of 4, no adapter scan, a tuning override).  The host feed asks it
 *     (quack_amd/host/pipeline.c); qk_accum_submit_fixed pads by itself on the way into the pinned slot.
 *   qk_accum_commit_padded: pinned-slot form; read r was written at r * stride of the acquired slot. */
int qk_accum_padded_stride(qk_accum *acc, uint32_t read_len, uint32_t *stride);
int qk_accum_commit_padded(qk_accum *acc, uint64_t n_reads, uint32_t read_len, uint32_t stride);
/* copying feed from caller-owned host memory (any size; split internally) */
int qk_accum_submit_strided(qk_accum *acc, const uint8_t *seq, const uint8_t *qual,
                            const uint32_t *lengths, uint32_t stride, uint64_t n_reads);

/* Wait for everything enqueued so far. */
int qk_accum_sync(qk_accum *acc);

/* Current sizes: max_length and number_of_sequences (quack.c:225-226). */
int qk_accum_stats(qk_accum *acc, uint64_t *max_len, uint64_t *n_reads);

/* Device-side table for collectives: planar u64 [QK_N_ROWS][table_len] + 1
 * trailing word (number_of_sequences).  Integer sums commute, so an
 * all-reduce(SUM) over these words from N accumulators that saw disjoint read
 * batches equals one accumulator that saw them all. */
int qk_accum_table_words(qk_accum *acc, uint64_t *n_words);
int qk_accum_reserve(qk_accum *acc, uint64_t max_len); /* grow to common size */
int qk_accum_export_table(qk_accum *acc, void *d_dst, void *hip_stream);
int qk_accum_import_table(qk_accum *acc, const void *d_src, uint64_t max_len,
                          void *hip_stream);

/* In-process multi-GPU: one accumulator per device, one RCCL all-reduce of
 * the integer tables over xGMI (ncclUint64, ncclSum).  After it every
 * accumulator holds the global table. */
int qk_accum_allreduce(qk_accum **accs, int n);

/* Synchronise and copy out: out[0..max_len) in reference layout. */
int qk_accum_finish(qk_accum *acc, qk_base_info *out, uint64_t cap_positions,
                    uint64_t *max_len, uint64_t *n_reads);

/* ---- timing hooks (bench.py / kbench) --------------------------------- */
/* Average duration in ms of the histogram kernel launches recorded since the
 * last reset, measured with hipEvents on the launch stream. */
/* on = N > 1: only every Nth batch is timed (the events themselves cost ~10 us of stream time per
 * batch; a caller that also takes the wall clock of the whole run samples) */
int qk_accum_timing_enable(qk_accum *acc, int on);
int qk_accum_timing_read(qk_accum *acc, double *total_ms, uint64_t *launches);
/* Same, plus the time of ALL kernels of the batches (reach pre-pass, length
 * kernel, first-hit reset, histogram, adapter count): hist_ms <= batch_ms. */
int qk_accum_timing_read_batch(qk_accum *acc, double *hist_ms, double *batch_ms,
                               uint64_t *launches);
/* Shortest and longest histogram-kernel launch among those (ms); both 0 when none was timed. */
int qk_accum_timing_read_range(qk_accum *acc, double *hist_min_ms, double *hist_max_ms);

/* ---- launch geometry of the caller's choosing: the experiment build (libquack_hip_exp.so) has the kernel variants for it, the
 * product library only those the planner itself picks (a launch under another geometry fails with QK_EINVAL there) ---------- */
int qk_accum_configure(qk_accum *acc, int threads_per_wg, int unroll,
                       int tile_positions, int wgs_per_cu);

/* ---- diagnostics (tests/test_planner.py, tools/kbench; no reference counterpart) -- */
/* Launch geometry the planner would choose for a batch shape; needs no device.
 * out[16]: n_tiles, tile_pos, chunks, reads/iter, unroll, pipe, reads/slice,
 * n_slices, n_blocks, LDS bytes, halo, fused, dynamic, aligned, replicas,
 * row dwords. */
int qk_debug_plan(uint64_t n_reads, uint32_t max_len, int ragged, int adapters,
                  uint32_t bucket_log2, int gapped, int aligned, int n_cu,
                  uint64_t *out);
/* How a fixed-length batch with the adapter scan is cut into rows of several reads (16-position lanes filled better:
 * 150 bp at stride 152 -> two reads per row); out[8]: group, row positions, tile_pos, rows/iter, unroll, w16,
 * bucket_log2, LDS bytes. */
int qk_debug_group(uint64_t n_reads, uint32_t read_len, uint32_t stride, uint32_t bucket_log2, uint64_t *out);
/* Where the wide LDS layout of the fixed-length adapter kernel (16 positions per lane) puts its pieces for `chunks` 8-position
 * chunks per row, `replicas` counter replicas, a bucket table of 2^bucket_log2 16-byte buckets (0: none) and a first-hit ring of
 * fh_words words — the function the planner and the kernel share (qk::wide_plan).  out[8]: planes, bucket table at (dwords from
 * LDS byte 0), ring at (0: in the unused column pairs of the last plane), dynamic LDS bytes (0: the shape does not fit), words
 * of ring the last plane has room for, dwords of the small rows' gap that are taken, 0, 0. */
int qk_debug_wide(uint32_t chunks, uint32_t replicas, uint32_t bucket_log2, uint32_t fh_words, uint64_t *out);
/* Ablation builds of the histogram kernel (-DQK_ABLATION, tools/kbench only);
 * the product library only has mode 0 and rejects any other at launch. */
int qk_debug_set_mode(int mode);

#ifdef __cplusplus
}
#endif
#endif /* QUACK_HIP_H */
