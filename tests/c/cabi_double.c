/*
 * cabi_double.c — TEST DOUBLE of include/quack_hip.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Lets the CPU tier run the real C host (cli.c, pipeline.c, reader.c, source.c,
 * render.c) end to end without a GPU: the part of the C-ABI the host calls is
 * implemented here on top of the test oracle, with deliberately tiny batch
 * slots so that the host's batching, slot alternation, gapped/aligned
 * switching, sharding and paired threads are all exercised.  It is linked only
 * into the test binary built by tests/test_host_pipeline_cpu.py; nothing under
 * quack_amd/ refers to it and the product has no CPU path
 * (tests/test_cabi_and_cli.py::test_product_does_not_reference_the_oracle).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "quack_hip.h"
#include "quack_oracle.h"

static _Thread_local char err[256];
static int fail(int code, const char *msg) {
  snprintf(err, sizeof err, "%s", msg);
  return code;
}
const char *qk_last_error(void) { return err; }
const char *qk_version(void) { return "C-ABI test double (oracle-backed)"; }
int qk_device_count(int *n) {
  *n = 4;
  return QK_OK;
}

struct qk_accum {
  oracle_table t;
  uint8_t *kmers;           /* oracle form: one byte per 10-mer */
  uint8_t *seq[2], *qual[2];
  uint64_t *off[2];
  uint32_t *len[2];
  int next, held;
  uint64_t cap_bytes, cap_reads;
  unsigned long commits, gapped_commits, aligned_commits, strided_commits, padded_commits, copied_submits, resizes, neutral_commits;
};

int qk_accum_create(qk_accum **out, int device, const uint32_t *bits, uint64_t hint) {
  (void)hint;
  if (device < 0 || device >= 4) return fail(QK_EINVAL, "device out of range");
  if (getenv("QK_DOUBLE_CREATE_DELAY_MS")) usleep(1000 * (useconds_t)atoi(getenv("QK_DOUBLE_CREATE_DELAY_MS")));   /* a slow HIP start-up */
  qk_accum *a = calloc(1, sizeof *a);
  const char *e = getenv("QK_DOUBLE_SLOT_BYTES");
  oracle_table_init(&a->t);
  a->cap_bytes = e ? strtoull(e, 0, 10) : 70000;   /* tiny on purpose */
  a->cap_reads = a->cap_bytes / 32 + 16;
  a->held = -1;
  for (int i = 0; i < 2; i++) {
    a->seq[i] = malloc(a->cap_bytes + QK_TAIL_SLACK);
    a->qual[i] = malloc(a->cap_bytes + QK_TAIL_SLACK);
    a->off[i] = malloc((a->cap_reads + 1) * sizeof(uint64_t));
    a->len[i] = malloc(a->cap_reads * sizeof(uint32_t));
  }
  if (bits) {
    a->kmers = calloc(ORACLE_KMER_TABLE, 1);
    for (uint32_t k = 0; k < ORACLE_KMER_TABLE; k++) a->kmers[k] = (bits[k >> 5] >> (k & 31)) & 1;
  }
  *out = a;
  return QK_OK;
}

void qk_accum_destroy(qk_accum *a) {
  if (!a) return;
  if (getenv("QK_DOUBLE_VERBOSE"))
    fprintf(stderr, "[double] commits %lu gapped %lu aligned %lu strided %lu copied %lu resized %lu padded %lu neutral %lu\n", a->commits, a->gapped_commits,
            a->aligned_commits, a->strided_commits, a->copied_submits, a->resizes, a->padded_commits, a->neutral_commits);
  for (int i = 0; i < 2; i++) {
    free(a->seq[i]);
    free(a->qual[i]);
    free(a->off[i]);
    free(a->len[i]);
  }
  free(a->kmers);
  oracle_table_free(&a->t);
  free(a);
}

int qk_accum_acquire(qk_accum *a, uint8_t **seq, uint8_t **qual, uint64_t **off, uint64_t *cb, uint64_t *cr) {
  if (a->held >= 0) return fail(QK_ESTATE, "a batch is already acquired");
  a->held = a->next;
  *seq = a->seq[a->held];
  *qual = a->qual[a->held];
  *off = a->off[a->held];
  if (cb) *cb = a->cap_bytes;
  if (cr) *cr = a->cap_reads;
  return QK_OK;
}

int qk_accum_resize_slots(qk_accum *a, uint64_t min_bytes) {
  if (a->held >= 0) return fail(QK_ESTATE, "a batch is acquired");
  if (getenv("QK_DOUBLE_NO_RESIZE")) return fail(QK_ENOMEM, "no memory for bigger slots");
  if (min_bytes <= a->cap_bytes) return QK_OK;
  a->cap_bytes = min_bytes;
  a->cap_reads = a->cap_bytes / 32 + 16;
  for (int i = 0; i < 2; i++) {
    free(a->seq[i]);
    free(a->qual[i]);
    free(a->off[i]);
    free(a->len[i]);
    a->seq[i] = malloc(a->cap_bytes + QK_TAIL_SLACK);
    a->qual[i] = malloc(a->cap_bytes + QK_TAIL_SLACK);
    a->off[i] = malloc((a->cap_reads + 1) * sizeof(uint64_t));
    a->len[i] = malloc(a->cap_reads * sizeof(uint32_t));
  }
  a->resizes++;
  return QK_OK;
}

int qk_accum_slot_lengths(qk_accum *a, uint32_t **lengths) {
  if (a->held < 0) return fail(QK_ESTATE, "no batch acquired");
  *lengths = a->len[a->held];
  return QK_OK;
}

static void release(qk_accum *a) {
  a->held = -1;
  a->next ^= 1;
}

int qk_accum_commit(qk_accum *a, uint64_t n, uint64_t total, int offsets_used, uint32_t read_len) {
  if (a->held < 0) return fail(QK_ESTATE, "no batch acquired");
  const int s = a->held;
  if (total > a->cap_bytes || n > a->cap_reads) return fail(QK_EINVAL, "batch exceeds slot capacity");
  if (offsets_used && (a->off[s][0] != 0 || a->off[s][n] != total)) return fail(QK_EINVAL, "bad offsets");
  if (!offsets_used && (uint64_t)read_len * n != total) return fail(QK_EINVAL, "bad fixed batch");
  release(a);
  if (n == 0) return QK_OK;   /* (an empty batch only touches the slot) */
  a->commits++;
  oracle_accumulate_batch(&a->t, a->seq[s], a->qual[s], offsets_used ? a->off[s] : NULL, n, read_len, a->kmers);
  return QK_OK;
}

/* the copying feed: caller-owned memory of any size */
int qk_accum_submit(qk_accum *a, const uint8_t *seq, const uint8_t *qual, const uint64_t *offsets, uint64_t n) {
  if (a->held >= 0) return fail(QK_ESTATE, "a batch is acquired");
  if (n && offsets[0] != 0) return fail(QK_EINVAL, "bad offsets");
  a->copied_submits++;
  oracle_accumulate_batch(&a->t, seq, qual, offsets, n, 0, a->kmers);
  return QK_OK;
}
int qk_accum_submit_fixed(qk_accum *a, const uint8_t *seq, const uint8_t *qual, uint32_t read_len, uint64_t n) {
  if (a->held >= 0) return fail(QK_ESTATE, "a batch is acquired");
  a->copied_submits++;
  oracle_accumulate_batch(&a->t, seq, qual, NULL, n, read_len, a->kmers);
  return QK_OK;
}

int qk_accum_commit_gapped(qk_accum *a, uint64_t n, uint64_t extent, uint32_t flags) {
  if (a->held < 0) return fail(QK_ESTATE, "no batch acquired");
  const int s = a->held;
  uint64_t end = 0;
  if (extent > a->cap_bytes || n > a->cap_reads) return fail(QK_EINVAL, "batch exceeds slot capacity");
  for (uint64_t i = 0; i < n; i++) {
    if (a->off[s][i] < end) return fail(QK_EINVAL, "reads overlap");
    if ((flags & QK_BATCH_ALIGNED128) && (a->off[s][i] & 127)) return fail(QK_EINVAL, "read is not 128-byte aligned");
    end = a->off[s][i] + a->len[s][i];
  }
  if (end > extent) return fail(QK_EINVAL, "reads end past the extent");
  release(a);
  a->commits++;
  a->gapped_commits++;
  a->aligned_commits += (flags & QK_BATCH_ALIGNED128) != 0;
  for (uint64_t i = 0; i < n; i++)
    oracle_accumulate_read(&a->t, a->seq[s] + a->off[s][i], a->qual[s] + a->off[s][i], a->len[s][i], a->kmers);
  return QK_OK;
}

int qk_accum_commit_strided_flags(qk_accum *a, uint64_t n, uint32_t stride, uint32_t flags) {
  if (a->held < 0) return fail(QK_ESTATE, "no batch acquired");
  const int s = a->held;
  if (stride == 0 || (stride & 3u)) return fail(QK_EINVAL, "stride must be a multiple of 4");
  if (flags & ~QK_BATCH_NEUTRAL_PADS) return fail(QK_EINVAL, "unknown flags");
  if (n * (uint64_t)stride > a->cap_bytes || n > a->cap_reads) return fail(QK_EINVAL, "batch exceeds slot capacity");
  for (uint64_t i = 0; i < n; i++) {
    if (a->len[s][i] > stride) return fail(QK_EINVAL, "read longer than the stride");
    /* the double holds the producer to its promise for EVERY pad byte */
    for (uint32_t k = a->len[s][i]; (flags & QK_BATCH_NEUTRAL_PADS) && k < stride; k++)
      if (a->seq[s][i * stride + k] != 0xFF || a->qual[s][i * stride + k] != 0xFF) return fail(QK_EINVAL, "a pad byte is not 0xFF");
  }
  a->neutral_commits += (flags & QK_BATCH_NEUTRAL_PADS) != 0;
  release(a);
  a->commits++;
  a->strided_commits++;
  for (uint64_t i = 0; i < n; i++)
    oracle_accumulate_read(&a->t, a->seq[s] + i * stride, a->qual[s] + i * stride, a->len[s][i], a->kmers);
  return QK_OK;
}

int qk_accum_commit_strided(qk_accum *a, uint64_t n, uint32_t stride) { return qk_accum_commit_strided_flags(a, n, stride, 0u); }

/* padded fixed-length batches: the double wants them for uniform reads whose length is not a multiple of 4 when
 * adapters are loaded (the product's rule), or for every such length with QK_DOUBLE_PAD_ALWAYS */
int qk_accum_padded_stride(qk_accum *a, uint32_t read_len, uint32_t *stride) {
  *stride = 0;
  if ((read_len & 3u) && read_len >= 16 && (a->kmers || getenv("QK_DOUBLE_PAD_ALWAYS"))) *stride = (read_len + 3u) & ~3u;
  return QK_OK;
}

int qk_accum_commit_padded(qk_accum *a, uint64_t n, uint32_t read_len, uint32_t stride) {
  if (a->held < 0) return fail(QK_ESTATE, "no batch acquired");
  const int s = a->held;
  if (stride == 0 || (stride & 3u) || read_len > stride) return fail(QK_EINVAL, "stride must be a multiple of 4 and >= read_len");
  if (n * (uint64_t)stride > a->cap_bytes || n > a->cap_reads) return fail(QK_EINVAL, "batch exceeds slot capacity");
  release(a);
  if (n == 0) return QK_OK;
  a->commits++;
  a->padded_commits++;
  for (uint64_t i = 0; i < n; i++)
    oracle_accumulate_read(&a->t, a->seq[s] + i * stride, a->qual[s] + i * stride, read_len, a->kmers);
  return QK_OK;
}

static void add_tables(oracle_table *dst, const oracle_table *src) {
  oracle_table sum;
  oracle_table_init(&sum);
  sum.max_length = dst->max_length > src->max_length ? dst->max_length : src->max_length;
  sum.capacity = sum.max_length ? sum.max_length : 1;
  sum.bases = calloc(sum.capacity * ORACLE_ROWS, sizeof(uint64_t));
  for (uint64_t i = 0; i < dst->max_length * ORACLE_ROWS; i++) sum.bases[i] += dst->bases[i];
  for (uint64_t i = 0; i < src->max_length * ORACLE_ROWS; i++) sum.bases[i] += src->bases[i];
  sum.number_of_sequences = dst->number_of_sequences + src->number_of_sequences;
  free(dst->bases);
  *dst = sum;
}

int qk_accum_allreduce(qk_accum **accs, int n) {
  for (int i = 1; i < n; i++) add_tables(&accs[0]->t, &accs[i]->t);
  for (int i = 1; i < n; i++) {   /* everybody ends up with the sum */
    oracle_table_free(&accs[i]->t);
    oracle_table_init(&accs[i]->t);
    add_tables(&accs[i]->t, &accs[0]->t);
  }
  return QK_OK;
}

int qk_accum_finish(qk_accum *a, qk_base_info *out, uint64_t cap, uint64_t *max_len, uint64_t *n_reads) {
  if (max_len) *max_len = a->t.max_length;
  if (n_reads) *n_reads = a->t.number_of_sequences;
  if (!out) return QK_OK;
  if (cap < a->t.max_length) return fail(QK_EINVAL, "output too small");
  memcpy(out, a->t.bases, a->t.max_length * ORACLE_ROWS * sizeof(uint64_t));
  return QK_OK;
}
