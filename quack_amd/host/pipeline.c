/*
 * pipeline.c — whole-file accumulation: tokenizer -> pinned batches -> C-ABI.
 *
 * This is read_fastq() (quack.c:180-228) with its loop body moved to the GPU:
 * the host thread parses records straight into the accumulator's pinned batch
 * slot (qk_accum_acquire), commits it (async H2D + kernels on that slot's
 * stream) and immediately parses the next batch into the other slot, so
 * inflate/parse, PCIe copy and kernels overlap.  With several devices the
 * batches are dealt round-robin and the integer tables are summed once with
 * RCCL (qk_accum_allreduce).  There is no CPU fallback: any C-ABI error aborts
 * the file.
 */
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "quack_host.h"

static _Thread_local char host_err[512];

const char *qkh_last_error(void) { return host_err; }

static int host_fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(host_err, sizeof host_err, fmt, ap);
  va_end(ap);
  return -1;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* QUACK_DEVICES=0,1,... -> device ids (default: device 0).  Environment only:
 * the command line stays the reference's (quack.c:59-132). */
int qkh_device_list(int *devs, int cap) {
  const char *env = getenv("QUACK_DEVICES");
  int n = 0;
  if (!env || !*env) {
    devs[0] = 0;
    return 1;
  }
  while (*env && n < cap) {
    char *end;
    long v = strtol(env, &end, 10);
    if (end == env) break;
    devs[n++] = (int)v;
    env = *end == ',' ? end + 1 : end;
  }
  if (n == 0) devs[n++] = 0;
  return n;
}

int qkh_accumulate_file(const char *path, const uint32_t *bitset,
                        const int *devices, int n_devices,
                        qk_base_info **bases_out, uint64_t *max_len,
                        uint64_t *n_reads) {
  qk_accum *accs[64];
  qkh_reader *rd = NULL;
  int rc = -1, turn = 0, made = 0, long_reads = 0;
  uint32_t stride = 0;   /* != 0: short reads of nearly one length, laid out at a fixed stride */
  const int no_stride = getenv("QUACK_NO_STRIDE") != NULL;
  const int verbose = getenv("QUACK_VERBOSE") != NULL;
  const double t0 = now_s();
  double t_created, t_first = 0, t_parsed;
  *bases_out = NULL;
  *max_len = *n_reads = 0;
  if (n_devices < 1 || n_devices > 64) return host_fail("bad device count %d", n_devices);
  /* the reader first: its producer threads inflate (and index) the first slices while the HIP runtime
   * starts up and the pinned slots are allocated — 0.2-0.3 s in which the tokenizer cannot run yet */
  rd = qkh_reader_open(path);
  if (!rd) {
    host_fail("cannot open %s", path);
    goto out;
  }
  for (; made < n_devices; made++)
    if (qk_accum_create(&accs[made], devices[made], bitset, 0)) {
      host_fail("device %d: %s", devices[made], qk_last_error());
      goto out;
    }
  t_created = now_s();
  while (!qkh_reader_done(rd)) {
    uint8_t *seq, *qual;
    uint64_t *offsets, cap_bytes, cap_reads, total = 0;
    uint32_t uniform = 0;
    int64_t n;
    qk_accum *acc = accs[turn];
    if (qk_accum_acquire(acc, &seq, &qual, &offsets, &cap_bytes, &cap_reads)) {
      host_fail("%s", qk_last_error());
      goto out;
    }
    if (!t_first) t_first = now_s();
    if (stride) {
      /* short reads, nearly all of one length (judged by the previous batch; what trimmed
       * Illumina runs look like): fixed stride + per-read lengths, so that the kernels keep
       * the addressing and the pipelined loop of a fixed-length batch (qk_accum_commit_strided) */
      uint32_t *lengths;
      uint64_t parked;
      if (qk_accum_slot_lengths(acc, &lengths)) {
        host_fail("%s", qk_last_error());
        goto out;
      }
      n = qkh_reader_fill_strided(rd, seq, qual, lengths, cap_bytes, cap_reads, stride, &uniform);
      if (n < 0) {
        host_fail("%s: out of memory while parsing", path);
        goto out;
      }
      /* all of one length after all, and the stride is that length: a plain fixed-length batch */
      if (uniform == stride ? qk_accum_commit(acc, (uint64_t)n, (uint64_t)n * stride, 0, stride)
                            : qk_accum_commit_strided(acc, (uint64_t)n, stride)) {
        host_fail("%s", qk_last_error());
        goto out;
      }
      /* a read longer than the stride waits: widen the stride if it is still a short read, else go back to
       * packed batches */
      parked = qkh_reader_parked_len(rd);
      if (parked > stride) stride = parked <= 512 ? (uint32_t)((parked + 3) & ~3ull) : 0;
      turn = (turn + 1) % n_devices;
      continue;
    }
    if (long_reads) {
      /* long reads (judged by the previous batch): every read starts on a 128-byte
       * cache line, so that the position tiles the kernels cut them into are whole
       * lines (QK_BATCH_ALIGNED128, quack_hip.h); costs < 64 bytes of padding per read */
      uint32_t *lengths;
      if (qk_accum_slot_lengths(acc, &lengths)) {
        host_fail("%s", qk_last_error());
        goto out;
      }
      n = qkh_reader_fill_gapped(rd, seq, qual, offsets, lengths, cap_bytes, cap_reads, 128, &total, &uniform);
    } else {
      n = qkh_reader_fill(rd, seq, qual, offsets, cap_bytes, cap_reads, &total, &uniform);
    }
    if (n < 0) {
      host_fail(n == -4 ? "%s: a read exceeds the batch size (raise QUACK_HIP_BATCH_MB)"
                        : "%s: out of memory while parsing", path);
      goto out;
    }
    if (long_reads ? qk_accum_commit_gapped(acc, (uint64_t)n, total, QK_BATCH_ALIGNED128)
                   : qk_accum_commit(acc, (uint64_t)n, total, uniform == 0, uniform)) {
      host_fail("%s", qk_last_error());
      goto out;
    }
    if (n > 0) long_reads = total / (uint64_t)n >= 1024 && !getenv("QUACK_NO_ALIGN");
    if (n > 0 && !long_reads && !uniform && !no_stride) {
      /* a ragged batch of short reads: is it "one length, some of them trimmed"?  (longest read <= 512:
       * one position tile; mean >= 3/4 of it: the padding stays below a third of the traffic) */
      uint64_t longest = 0;
      for (int64_t i = 0; i < n; i++)
        if (offsets[i + 1] - offsets[i] > longest) longest = offsets[i + 1] - offsets[i];
      if (longest >= 16 && longest <= 512 && total * 4 >= (uint64_t)n * longest * 3) stride = (uint32_t)((longest + 3) & ~3ull);
    }
    turn = (turn + 1) % n_devices;
  }
  t_parsed = now_s();
  if (n_devices > 1 && qk_accum_allreduce(accs, n_devices)) {
    host_fail("%s", qk_last_error());
    goto out;
  }
  if (qk_accum_finish(accs[0], NULL, 0, max_len, n_reads)) {
    host_fail("%s", qk_last_error());
    goto out;
  }
  if (*max_len) {
    *bases_out = calloc(*max_len, sizeof(qk_base_info));
    if (!*bases_out) {
      host_fail("out of memory");
      goto out;
    }
    if (qk_accum_finish(accs[0], *bases_out, *max_len, max_len, n_reads)) {
      host_fail("%s", qk_last_error());
      free(*bases_out);
      *bases_out = NULL;
      goto out;
    }
  }
  if (verbose)
    fprintf(stderr, "[quack] %s: accumulators %.3f s, first slot %.3f s, parse+submit %.3f s, drain+finish %.3f s\n",
            path, t_created - t0, t_first - t_created, t_parsed - t_first, now_s() - t_parsed);
  rc = 0;
out:
  if (rd) qkh_reader_close(rd);
  while (made-- > 0) qk_accum_destroy(accs[made]);
  return rc;
}

int qkh_render_document(FILE *out, FILE *err, const char *name, int adapters,
                        qk_base_info *fwd, uint64_t fwd_max_len, uint64_t fwd_reads,
                        qk_base_info *rev, uint64_t rev_max_len, uint64_t rev_reads) {
  qkh_svg w;
  uint64_t original;
  if (!fwd || fwd_max_len == 0 || (rev && rev_max_len == 0)) return -1;
  qkh_svg_begin(&w, out, rev != NULL, adapters, name);
  qkh_transform(fwd, &fwd_max_len, &original, fwd_reads, err);
  qkh_draw(&w, fwd, fwd_max_len, fwd_reads, 0, adapters);
  if (rev) {
    qkh_transform(rev, &rev_max_len, &original, rev_reads, err);
    qkh_draw(&w, rev, rev_max_len, rev_reads, 1, adapters);
  }
  qkh_svg_end(&w);
  return 0;
}
