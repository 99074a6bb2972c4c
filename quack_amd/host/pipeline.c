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
#include <stdatomic.h>
#include <sys/mman.h>
#include <pthread.h>
#include <sched.h>
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

static int g_process_exits;
void qkh_process_exits_after_this(int on) { g_process_exits = on; }
int qkh_process_exits(void) { return g_process_exits; }

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

/* ---- start-up overlap -------------------------------------------------------
 * Creating an accumulator (HIP runtime start-up, module load, device buffers: 0.16-0.3 s) and
 * the first use of its pinned slots (0.03 s each) happen on a thread of their own, while the
 * tokenizer already parses the first reads into ordinary heap batches ("early" batches, 16 MiB
 * per array).  Round 5: that thread, once the accumulators exist, also carries the early batches
 * into the pinned slots and commits them — a queue between the two threads, its buffers reused
 * — while the tokenizer goes on parsing; when the queue runs empty the tokenizer takes the pinned
 * slots over and fills them directly.  (Rounds 3-4 parsed at most 32 early batches — a third of
 * config 2's file: the tokenizer stood still whenever the runtime took longer than 0.12 s to
 * come up — and submitted them before parsing on: 0.045 s more.) */
enum { EARLY_BYTES = 16 << 20, EARLY_READS = 1 << 20, EARLY_MAX = 48 };   /* at most 1.5 GiB of reads + 0.4 GiB of offsets in flight */
typedef struct {
  uint8_t *seq, *qual;
  uint64_t *off;
  int64_t n;
  uint64_t total;
  uint32_t uniform;
} early_batch;

typedef struct {
  qk_accum **accs;
  const int *devices;
  int n_devices, made, failed;
  const uint32_t *bitset;
  char err[300];
  atomic_int done;          /* the accumulators exist (or could not be made: failed) */
  /* the queue of early batches: the tokenizer publishes `head`, this thread `tail`; batch i lives in early[i % EARLY_MAX] */
  early_batch *early;
  atomic_uint head, tail;
  atomic_int stop;          /* the tokenizer parses no more early batches */
  int no_stride, turn, submitted;
  uint32_t stride;          /* the padded stride the library asked for (0: none), for the batches behind the early ones */
  double t_created, t_busy;
} acc_creator;

static int submit_early(acc_creator *c, early_batch *e);

static void *acc_creator_main(void *p) {
  acc_creator *c = p;
  for (; c->made < c->n_devices; c->made++) {
    uint8_t *seq, *qual;
    uint64_t *offsets;
    qk_accum *a;
    if (qk_accum_create(&c->accs[c->made], c->devices[c->made], c->bitset, 0)) {
      snprintf(c->err, sizeof c->err, "device %d: %s", c->devices[c->made], qk_last_error());
      c->failed = 1;
      break;
    }
    a = c->accs[c->made];
    /* touch both pinned slots (allocated on first use): two empty batches */
    for (int k = 0; k < 2 && !c->failed; k++)
      if (qk_accum_acquire(a, &seq, &qual, &offsets, NULL, NULL) || qk_accum_commit(a, 0, 0, 0, 0)) {
        snprintf(c->err, sizeof c->err, "device %d: %s", c->devices[c->made], qk_last_error());
        c->failed = 1;
      }
    if (c->failed) {
      c->made++;   /* created: the caller destroys it */
      break;
    }
  }
  c->t_created = now_s();
  atomic_store(&c->done, 1);
  /* the early batches, in the order they were parsed, until the tokenizer says there will be no more */
  for (unsigned idle = 0;;) {
    const unsigned t = atomic_load(&c->tail);
    if (t != atomic_load(&c->head)) {
      const double t0 = now_s();
      if (!c->failed && submit_early(c, &c->early[t % EARLY_MAX])) {
        snprintf(c->err, sizeof c->err, "%.290s", host_err);
        c->failed = 1;   /* (the queue is still emptied: the tokenizer must not wait for a slot for ever) */
      }
      c->t_busy += now_s() - t0;
      c->submitted++;
      atomic_store(&c->tail, t + 1u);
      idle = 0;
      continue;
    }
    if (atomic_load(&c->stop) && t == atomic_load(&c->head)) break;   /* (head is final once stop is set) */
    if (++idle < 200) sched_yield();
    else {
      struct timespec ts = {0, 50000};
      nanosleep(&ts, NULL);
    }
  }
  return NULL;
}

/* memcpy on several threads (an early batch into a pinned slot: 40 MB, ~10 GB/s on one core) */
typedef struct {
  void *dst;
  const void *src;
  size_t n;
} copy_job;
static void *copy_main(void *p) {
  const copy_job *j = p;
  memcpy(j->dst, j->src, j->n);
  return NULL;
}
enum { COPY_THREADS = 8 };
static void parallel_copy(copy_job *jobs, int n_jobs) {
  /* cut the jobs into pieces of equal size, one thread per piece (this thread takes the last) */
  copy_job piece[COPY_THREADS];
  pthread_t th[COPY_THREADS];
  size_t all = 0, share;
  int n_piece = 0, started = 0;
  for (int i = 0; i < n_jobs; i++) all += jobs[i].n;
  share = all / COPY_THREADS + 4096;
  for (int i = 0; i < n_jobs; i++) {
    size_t at = 0;
    while (at < jobs[i].n) {
      size_t len = jobs[i].n - at < share ? jobs[i].n - at : share;
      if (n_piece == COPY_THREADS) {   /* (rounding: the rest goes with the last piece's thread) */
        memcpy((char *)jobs[i].dst + at, (const char *)jobs[i].src + at, jobs[i].n - at);
        break;
      }
      piece[n_piece].dst = (char *)jobs[i].dst + at;
      piece[n_piece].src = (const char *)jobs[i].src + at;
      piece[n_piece].n = len;
      n_piece++;
      at += len;
    }
  }
  for (int i = 0; i + 1 < n_piece; i++) {
    if (pthread_create(&th[i], NULL, copy_main, &piece[i])) break;
    started++;
  }
  for (int i = started; i < n_piece; i++) copy_main(&piece[i]);
  for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
}

/* an early batch of uniform reads into a pinned slot at a padded stride (qk_accum_commit_padded), on the same threads */
typedef struct {
  uint8_t *dst_seq, *dst_qual;
  const uint8_t *src_seq, *src_qual;
  uint64_t r0, r1;
  uint32_t len, stride;
} pad_job;
static void *pad_main(void *p) {
  const pad_job *j = p;
  for (uint64_t r = j->r0; r < j->r1; r++) {
    memcpy(j->dst_seq + r * j->stride, j->src_seq + r * j->len, j->len);
    memcpy(j->dst_qual + r * j->stride, j->src_qual + r * j->len, j->len);
    memset(j->dst_seq + r * j->stride + j->len, 0, j->stride - j->len);   /* (never counted; kept defined) */
    memset(j->dst_qual + r * j->stride + j->len, 0, j->stride - j->len);
  }
  return NULL;
}
static void parallel_pad(uint8_t *dst_seq, uint8_t *dst_qual, const uint8_t *src_seq, const uint8_t *src_qual, uint64_t n,
                         uint32_t len, uint32_t stride) {
  pad_job job[COPY_THREADS];
  pthread_t th[COPY_THREADS];
  int started = 0;
  for (int i = 0; i < COPY_THREADS; i++) {
    job[i] = (pad_job){dst_seq, dst_qual, src_seq, src_qual, n * (uint64_t)i / COPY_THREADS, n * (uint64_t)(i + 1) / COPY_THREADS, len, stride};
  }
  for (int i = 0; i + 1 < COPY_THREADS; i++) {
    if (pthread_create(&th[i], NULL, pad_main, &job[i])) break;
    started++;
  }
  for (int i = started; i < COPY_THREADS; i++) pad_main(&job[i]);
  for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
}

/* heap for an early batch: 2 MiB aligned and marked for transparent huge pages where the kernel
 * offers them (first-touch faults and the later munmap are per page: 40 MB = 10,000 small ones) */
static void *early_alloc(size_t n) {
  void *p = NULL;
  n = (n + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
  if (posix_memalign(&p, 2u << 20, n)) return NULL;
#ifdef MADV_HUGEPAGE
  (void)madvise(p, n, MADV_HUGEPAGE);
#endif
  return p;
}

static void early_free(early_batch *e) {
  free(e->seq);
  free(e->qual);
  free(e->off);
  memset(e, 0, sizeof *e);
}
static void *early_reaper_main(void *p) {
  early_batch *early = p;
  for (int i = 0; i < EARLY_MAX; i++) early_free(&early[i]);
  return NULL;
}

/* one early batch into the next accumulator's pinned slot (or, when the slots are smaller than an early batch, through the
 * copying feed); runs on the creator thread */
static int submit_early(acc_creator *c, early_batch *e) {
  qk_accum *acc = c->accs[c->turn];
  uint8_t *seq, *qual;
  uint64_t *offsets, cap_bytes, cap_reads;
  uint32_t pad = 0;
  if (qk_accum_acquire(acc, &seq, &qual, &offsets, &cap_bytes, &cap_reads)) return host_fail("%s", qk_last_error());
  if (e->uniform && !c->no_stride && (qk_accum_padded_stride(acc, e->uniform, &pad) || pad <= e->uniform)) pad = 0;
  if (pad && (uint64_t)e->n * pad <= cap_bytes && (uint64_t)e->n <= cap_reads) {
    /* uniform reads the library wants at a padded stride (see below): re-laid on the way into the slot */
    parallel_pad(seq, qual, e->seq, e->qual, (uint64_t)e->n, e->uniform, pad);
    if (qk_accum_commit_padded(acc, (uint64_t)e->n, e->uniform, pad)) return host_fail("%s", qk_last_error());
    if (pad <= 512) c->stride = pad;   /* the batches behind the early ones are parsed into that layout directly */
  } else if (e->total <= cap_bytes && (uint64_t)e->n <= cap_reads) {
    copy_job jobs[3] = {{seq, e->seq, e->total}, {qual, e->qual, e->total},
                        {offsets, e->off, e->uniform ? 0 : ((size_t)e->n + 1) * sizeof(uint64_t)}};
    parallel_copy(jobs, 3);
    if (qk_accum_commit(acc, (uint64_t)e->n, e->total, e->uniform == 0, e->uniform)) return host_fail("%s", qk_last_error());
  } else if (qk_accum_commit(acc, 0, 0, 0, 0) ||   /* slots smaller than an early batch: the copying feed splits it */
             (e->uniform ? qk_accum_submit_fixed(acc, e->seq, e->qual, e->uniform, (uint64_t)e->n)
                         : qk_accum_submit(acc, e->seq, e->qual, e->off, (uint64_t)e->n))) {
    return host_fail("%s", qk_last_error());
  }
  c->turn = (c->turn + 1) % c->n_devices;
  return 0;
}

int qkh_accumulate_file(const char *path, const uint32_t *bitset,
                        const int *devices, int n_devices,
                        qk_base_info **bases_out, uint64_t *max_len,
                        uint64_t *n_reads) {
  qk_accum *accs[64];
  qkh_reader *rd = NULL;
  int rc = -1, turn = 0, made = 0, long_reads = 0, n_early_made = 0, early_used = 0;
  early_batch early[EARLY_MAX];
  pthread_t reaper;
  int reaping = 0;
  memset(early, 0, sizeof early);
  uint32_t stride = 0;   /* != 0: short reads of nearly one length, laid out at a fixed stride */
  const int no_stride = getenv("QUACK_NO_STRIDE") != NULL;
  /* Short reads of mixed lengths go out at a fixed stride (0xFF behind every read) rather than packed when their mean length is at
   * least this many percent of the longest read.  Round 5 measured the two layouts on the same 10M reads (tools/ragged_probe.py,
   * profiles/r05_ragged_probe.log): the strided kernels take n x stride positions whatever the lengths — U[30,150]: 0.534 ms,
   * with adapters 0.596 — the packed (ragged) ones pay per read and per base — 0.559 / 0.914 ms at a fill of 0.60, 0.588 / 0.938 at
   * 0.83: with the adapter scan the stride wins by a third at every fill that was tried.  (Rounds 2-4: 75 % either way.)  What it
   * costs is PCIe traffic, of which the feed uses a sixth. */
  const unsigned stride_in = bitset ? 35u : 50u, stride_out = bitset ? 30u : 45u;
  const int verbose = getenv("QUACK_VERBOSE") != NULL;
  const double t0 = now_s();
  double t_created, t_early, t_first = 0, t_parsed, t_copying = 0;
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
  {
    acc_creator cr;
    pthread_t th;
    int threaded;
    memset(&cr, 0, sizeof cr);
    cr.accs = accs;
    cr.devices = devices;
    cr.n_devices = n_devices;
    cr.bitset = bitset;
    cr.early = early;
    cr.no_stride = no_stride;
    atomic_init(&cr.done, 0);
    atomic_init(&cr.head, 0);
    atomic_init(&cr.tail, 0);
    atomic_init(&cr.stop, 0);
    /* (tests shrink the early batches to meet the "read longer than an early batch" case) */
    const size_t early_bytes = getenv("QUACK_EARLY_BYTES") ? (size_t)strtoull(getenv("QUACK_EARLY_BYTES"), NULL, 10) : (size_t)EARLY_BYTES;
    unsigned head = 0;
    int parse_failed = 0;
    threaded = !getenv("QUACK_NO_EARLY") && pthread_create(&th, NULL, acc_creator_main, &cr) == 0;
    if (!threaded) {
      atomic_store(&cr.stop, 1);
      acc_creator_main(&cr);
    }
    while (threaded && !qkh_reader_done(rd)) {
      /* the accumulators exist and the other thread is down to the batch that was published a moment ago: from here on
       * straight into the pinned slots (the join below waits for that last copy: ~1 ms) */
      if (atomic_load(&cr.done) && (cr.failed || head - atomic_load(&cr.tail) <= 1u)) break;
      if (head - atomic_load(&cr.tail) == (unsigned)EARLY_MAX) {   /* every buffer is waiting for its turn */
        struct timespec ts = {0, 100000};
        nanosleep(&ts, NULL);
        continue;
      }
      early_batch *e = &early[head % EARLY_MAX];
      if (!e->seq) {
        e->seq = early_alloc(early_bytes + QK_TAIL_SLACK);
        e->qual = early_alloc(early_bytes + QK_TAIL_SLACK);
        e->off = early_alloc(((size_t)EARLY_READS + 1) * sizeof(uint64_t));
        early_used++;
        if (!e->seq || !e->qual || !e->off) {
          early_free(e);
          break;   /* no memory to spare: wait for the accumulators instead */
        }
      }
      e->n = qkh_reader_fill(rd, e->seq, e->qual, e->off, early_bytes, EARLY_READS, &e->total, &e->uniform);
      if (e->n <= 0) {
        if (e->n == -4) break;   /* a read longer than an early batch: it stays parked for a pinned slot */
        if (e->n == 0) continue;
        parse_failed = 1;
        break;
      }
      atomic_store(&cr.head, ++head);
    }
    if (threaded) {
      atomic_store(&cr.stop, 1);
      pthread_join(th, NULL);
    }
    made = cr.made;
    if (parse_failed) {
      host_fail("%s: out of memory while parsing", path);
      goto out;
    }
    if (cr.failed) {
      host_fail("%s", cr.err);
      goto out;
    }
    turn = cr.turn;
    stride = cr.stride;
    n_early_made = cr.submitted;
    t_created = cr.t_created;
    t_copying = cr.t_busy;
  }
  /* unmapping 1 GiB costs ~0.1 s (TLB shootdowns across the producer threads): off the critical path */
  if (early_used && pthread_create(&reaper, NULL, early_reaper_main, early) == 0) reaping = 1;
  else early_reaper_main(early);
  t_early = now_s();
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
      /* all of one length after all: a fixed-length batch — plain when the stride is that length, padded
       * (qk_accum_commit_padded: uniform reads whose length is not a multiple of 4) when it is the next multiple of 4 */
      if (uniform ? qk_accum_commit_padded(acc, (uint64_t)n, uniform, stride)
                  : qk_accum_commit_strided_flags(acc, (uint64_t)n, stride, QK_BATCH_NEUTRAL_PADS)) {   /* (the fill wrote 0xFF pads) */
        host_fail("%s", qk_last_error());
        goto out;
      }
      /* a read longer than the stride waits: widen the stride if it is still a short read, else go back to
       * packed batches */
      parked = qkh_reader_parked_len(rd);
      if (parked > stride) stride = parked <= 512 ? (uint32_t)((parked + 3) & ~3ull) : 0;
      else if (n > 0 && uniform) {
        /* (a stride that one long read widened earlier comes back down once the batches are uniform again) */
        if (((uniform + 3u) & ~3u) < stride) stride = (uniform + 3u) & ~3u;
      } else if (n > 0 && !uniform) {
        /* (the reads have become too short for the stride: see stride_in / stride_out) */
        uint64_t sum = 0;
        for (int64_t i = 0; i < n; i++) sum += lengths[i];
        if (sum * 100 < (uint64_t)n * stride * stride_out) stride = 0;
      }
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
    if (n == -4 && qkh_reader_parked_len(rd) > 0) {
      /* one read longer than a slot (it is parked on the heap): give this accumulator slots that hold it
       * with room to spare, and try again */
      const uint64_t want = qkh_reader_parked_len(rd) + qkh_reader_parked_len(rd) / 4 + 4096;
      if (qk_accum_commit(acc, 0, 0, 0, 0) || qk_accum_resize_slots(acc, want)) {
        host_fail("%s: a read of %llu bases does not fit a batch: %s", path,
                  (unsigned long long)qkh_reader_parked_len(rd), qk_last_error());
        goto out;
      }
      continue;
    }
    if (n < 0) {
      host_fail(n == -4 ? "%s: a read exceeds the batch size"
                        : "%s: out of memory while parsing", path);
      goto out;
    }
    if (long_reads ? qk_accum_commit_gapped(acc, (uint64_t)n, total, QK_BATCH_ALIGNED128)
                   : qk_accum_commit(acc, (uint64_t)n, total, uniform == 0, uniform)) {
      host_fail("%s", qk_last_error());
      goto out;
    }
    if (n > 0) long_reads = total / (uint64_t)n >= 1024 && !getenv("QUACK_NO_ALIGN");
    if (n > 0 && !long_reads && uniform && !no_stride) {
      /* uniform reads: does the library want them at a padded stride (a length that is not a multiple of 4 with the
       * adapter scan: dword-aligned chunks, 16 positions per lane)?  The following batches are then laid out that way. */
      uint32_t want = 0;
      if (qk_accum_padded_stride(acc, uniform, &want) == 0 && want > uniform && want <= 512) stride = want;
    }
    if (n > 0 && !long_reads && !uniform && !no_stride) {
      /* a ragged batch of short reads: trimmed reads (longest read <= 512: one position tile; mean length at least stride_in
       * percent of it) go out at a fixed stride from the next batch on */
      uint64_t longest = 0;
      for (int64_t i = 0; i < n; i++)
        if (offsets[i + 1] - offsets[i] > longest) longest = offsets[i + 1] - offsets[i];
      if (longest >= 16 && longest <= 512 && total * 100 >= (uint64_t)n * longest * stride_in) stride = (uint32_t)((longest + 3) & ~3ull);
    }
    turn = (turn + 1) % n_devices;
  }
  t_parsed = now_s();
  if (qkh_reader_failed(rd)) {
    host_fail("%s: out of memory while reading", path);
    goto out;
  }
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
  if (!t_first) t_first = t_parsed;
  if (verbose)
    fprintf(stderr, "[quack] %s: accumulators %.3f s; %d early batches (heap) until %.3f s, their copies + commits took %.3f s of the other thread; "
            "first slot %.3f s, parse+submit %.3f s, drain+finish %.3f s\n",
            path, t_created - t0, n_early_made, t_early - t0, t_copying, t_first - t_early, t_parsed - t_first, now_s() - t_parsed);
  rc = 0;
out:
  {
    const double t_out = now_s();
    if (reaping) pthread_join(reaper, NULL);
    else early_reaper_main(early);
    /* the CLI exits right after printing: pinned memory, device buffers, streams and the HIP runtime
     * itself are left to the operating system (0.16 s of a 0.8 s run otherwise) */
    if (!(g_process_exits && rc == 0)) {
      if (rd) qkh_reader_close(rd);
      while (made-- > 0) qk_accum_destroy(accs[made]);
    }
    if (verbose) fprintf(stderr, "[quack] %s: close + destroy %.3f s\n", path, now_s() - t_out);
  }
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
