/*
 * source.c — where the tokenizer's bytes come from.
 *
 * Decompressed bytes arrive in 4 MiB blocks, in stream order, from producer
 * threads, so that inflate (78 % of the reference's wall time)
 * overlaps parsing, the PCIe copies and the kernels.  Five producers:
 *
 *   bgzf     a BGZF file (gzip members of <= 64 KiB that carry their own size
 *            in a "BC" extra field — bgzip, htslib): a dispatcher walks the
 *            member headers and a pool of workers decodes runs of members in
 *            parallel, each straight into its ring block;
 *   pgzip    any other regular gzip file is memory-mapped and decoded by a
 *            pool of threads that start speculatively in the middle of the
 *            stream (pinflate.c);
 *   inflate  the same with one thread (small files, QUACK_THREADS=1,
 *            QUACK_NO_PGZIP=1): inflate_fast.c, ~1.65x zlib on FASTQ; every
 *            block carries the previous 32 KiB of output in front of its data,
 *            which is all DEFLATE can refer back to;
 *   zlib     gzip from a pipe, or QUACK_ZLIB=1: zlib's gzread, like the
 *            reference (quack.c:160,187);
 *   plain    a regular file that is not gzip: read(2).
 *
 * Whatever a producer cannot decode ends the stream at that byte — the
 * reference's read loop also just stops when gzread fails (quack.c:193).
 *
 * Member trailers.  The own decoders log the trailers they meet and the
 * producers compute the CRC-32 of the pieces between them (on their threads);
 * qkh_source_next chains the pieces (crc32_combine) and compares at every member
 * end.  On a mismatch (CRC or length) the stream ends where the REFERENCE's
 * would.  The reference reads through kseq, i.e. gzread(fp, buf, 16384) calls,
 * and zlib's gzread returns nothing at all from the call in which inflate meets
 * a bad trailer — so the stream ends at the START of that call, a multiple of
 * 16384 (goldens of the reference binary: tests/golden/svg/badcrc_*).  Which
 * call that is takes a small model of gzread (gzread_model below): after a
 * member boundary zlib decodes up to 16 KiB AHEAD into its own buffer, so the
 * bad trailer can be met up to 32 KiB of output before the caller gets there.
 * To be able to take those bytes back, the last 32 KiB of every chunk are
 * handed out together with the next chunk.
 */
#define _GNU_SOURCE   /* sched_getaffinity, CPU_COUNT */
#include "source.h"

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "inflate_fast.h"
#include "pinflate.h"

enum { BLOCK_BYTES = 4 << 20, HIST = 32768, SERIAL_RING = 4, MAX_WORKERS = 32, GZREAD_CHUNK = 16384,
       HOLD_BACK = 2 * GZREAD_CHUNK };

/* zlib's gzread (1.2.x, gzread.c: gz_read / gz_fetch / gz_decomp) under kseq's fixed 16384-byte calls,
 * as far as it decides WHERE a stream with a damaged member ends.  Offsets are absolute output bytes.
 *   call_start  where the current gzread call began (every completed call returned 16384 bytes)
 *   cpos        what the current call has been given so far
 *   dec         how far inflate has decoded: [cpos, dec) sits in zlib's output buffer
 *   look        the previous member ended: the next step looks for a header and decodes into the buffer
 * gz_decomp loses the output of the call that fails and gz_read then returns 0 for the whole gzread, so
 * the reference's stream ends at call_start. */
typedef struct {
  uint64_t call_start, cpos, dec;
  int look;
} gzread_model;

/* advance the model to the end of the member that ends at absolute offset `end`; if its trailer is bad,
 * return 1 with *cut = where the reference's stream ends */
static int gzread_model_member(gzread_model *m, uint64_t end, int bad, uint64_t *cut) {
  for (;;) {
    const uint64_t need = m->call_start + GZREAD_CHUNK - m->cpos;   /* > 0 */
    if (m->dec > m->cpos) {   /* buffered output first */
      const uint64_t take = m->dec - m->cpos < need ? m->dec - m->cpos : need;
      m->cpos += take;
      if (m->cpos == m->call_start + GZREAD_CHUNK) m->call_start = m->cpos;
      continue;
    }
    /* one gz_decomp: into zlib's buffer (2 x 8192) after a member boundary or for a short remainder,
     * else straight into the caller's 16384 bytes */
    const int direct = !m->look && need == GZREAD_CHUNK;
    const uint64_t room = direct ? need : GZREAD_CHUNK;
    const uint64_t produce = end - m->dec < room ? end - m->dec : room;
    m->look = 0;
    m->dec += produce;
    if (direct) {
      m->cpos += produce;
      if (m->cpos == m->call_start + GZREAD_CHUNK && m->dec < end) m->call_start = m->cpos;   /* (at the end: below) */
    }
    if (m->dec == end) {
      /* inflate reached the trailer in this very call — also when the output filled up exactly at the
       * member's end: the end-of-block code and the trailer need no output room */
      if (bad) {
        *cut = m->call_start;
        return 1;
      }
      m->look = 1;
      if (direct && m->cpos == m->call_start + GZREAD_CHUNK) m->call_start = m->cpos;
      return 0;
    }
  }
}

typedef struct {
  uint8_t *base;        /* allocation: HIST bytes of history + BLOCK_BYTES of data */
  uint8_t *data;        /* base + HIST */
  size_t len;
  int ready;            /* filled; the consumer may take it */
  /* bgzf task */
  const uint8_t *in;
  size_t in_len, expect;
  uint64_t file_off;    /* plain pool: the block is file bytes [file_off, file_off + in_len) (in != NULL marks a worker's block) */
  qkh_end_list el;      /* member ends inside the block, CRC-32 of the pieces between them */
  uint32_t *nl;         /* newline offsets of the block (qkh_index_lines), by its producer */
  size_t n_nl, cap_nl;
} block;

struct qkh_source {
  pthread_mutex_t mu;
  pthread_cond_t space, filled, work;
  block *ring;
  unsigned n_ring;
  unsigned head;        /* blocks assigned to a producer so far */
  unsigned tail;        /* blocks consumed so far */
  unsigned next_work;   /* bgzf: next assigned block without a worker */
  int holding;          /* the consumer still holds block tail */
  int done;             /* no further block will be assigned */
  int ended;            /* the consumer met an undecodable run: stream over */
  int stop;             /* closing */
  int n_threads;
  pthread_t threads[MAX_WORKERS + 1];
  char kind[32];
  /* inputs */
  gzFile gz;
  int fd;
  const uint8_t *map;
  size_t map_len;
  /* parallel inflate of one gzip stream: has its own slots and threads */
  qkh_pinflate *pz;
  /* serial inflate_fast */
  qkh_inflate *zf;
  uint8_t *hist;
  size_t hist_len;
  /* bgzf */
  qkh_inflate *worker_z[MAX_WORKERS];
  int n_workers;
  /* member CRC chain and the hold-back of qkh_source_next (own decoders only) */
  int checks;               /* this producer's members are checked here (not zlib / plain) */
  uint32_t run_crc;         /* CRC-32 of the current member so far */
  uint64_t run_len;         /* ... and its length */
  uint64_t delivered_raw;   /* bytes taken from the producers so far (absolute stream offset of the next chunk) */
  gzread_model model;
  uint8_t held[HOLD_BACK];
  size_t n_held;            /* undelivered tail of the previous chunk */
  const uint8_t *cur;       /* the chunk handed out by the producer layer, and how much of it was given out */
  size_t cur_len;
  int crc_ended;            /* a trailer did not match: the stream is over */
  /* line index of the bytes handed out (front .. front + cur_len), offsets relative to `front`:
   * two buffers, the new one is put together from the tail of the old one and the chunk's own */
  uint32_t *lines[2];
  size_t n_lines[2], cap_lines[2];
  int cur_lines;            /* which of the two describes the current chunk */
  int have_lines;           /* ... and whether it is complete (every chunk so far came with an index) */
  const uint32_t *plain_nl; /* no trailer checks (plain files): the current block's own index */
  size_t plain_n_nl;
  int oom;                  /* a producer thread could not allocate: see producer_oom */
  int plain_pool;           /* uncompressed regular file: workers pread() the blocks (and index their lines) */
  uint64_t file_len;
  size_t given;             /* bytes of the current chunk handed to the caller */
};

/* A producer ran out of memory: the stream ends where it stands, and qkh_source_failed() tells the reader
 * that this end is not the file's (the run must fail, not report the counts of a truncated stream). */
static size_t producer_oom(qkh_source *s) {
  __atomic_store_n(&s->oom, 1, __ATOMIC_RELAXED);
  return 0;
}

/* ------------------------------------------------------------- ring basics */
static block *claim_block(qkh_source *s) { /* producer side; NULL when closing */
  block *b;
  pthread_mutex_lock(&s->mu);
  while (!s->stop && s->head - s->tail == s->n_ring) pthread_cond_wait(&s->space, &s->mu);
  if (s->stop) {
    pthread_mutex_unlock(&s->mu);
    return NULL;
  }
  b = &s->ring[s->head % s->n_ring];
  b->ready = 0;
  b->len = 0;
  b->in = NULL;
  b->el.n = 0;
  pthread_mutex_unlock(&s->mu);
  return b;
}

static void finish_stream(qkh_source *s) {
  pthread_mutex_lock(&s->mu);
  s->done = 1;
  pthread_cond_broadcast(&s->filled);
  pthread_cond_broadcast(&s->work);
  pthread_mutex_unlock(&s->mu);
}

/* ---------------------------------------------------------- serial producer */
static size_t fill_serial(qkh_source *s, block *b) {
  size_t got = 0;
  if (s->zf) {
    long k = 1;
    memcpy(b->data - s->hist_len, s->hist, s->hist_len);
    b->el.n = 0;
    while (got < BLOCK_BYTES) {
      k = qkh_inflate_read(s->zf, b->data + got, BLOCK_BYTES - got, s->hist_len + got);
      /* (also the trailers met by a call that produced nothing) */
      if (qkh_end_list_take(&b->el, s->zf, got)) return producer_oom(s);
      if (k > 0) got += (size_t)k;
      else if (!(k == 0 && qkh_inflate_log_full(s->zf))) break;
    }
    if (qkh_end_list_crcs(&b->el, b->data, got)) return producer_oom(s);
    b->n_nl = qkh_index_lines(b->data, got, &b->nl, &b->cap_nl);
    if (got >= HIST) {
      memcpy(s->hist, b->data + got - HIST, HIST);
      s->hist_len = HIST;
    } else if (got) {
      const size_t keep = s->hist_len + got > HIST ? HIST - got : s->hist_len;
      memmove(s->hist, s->hist + s->hist_len - keep, keep);
      memcpy(s->hist + keep, b->data, got);
      s->hist_len = keep + got;
    }
  } else if (s->fd >= 0) {
    long n = 1;
    while (got < BLOCK_BYTES && (n = read(s->fd, b->data + got, BLOCK_BYTES - got)) > 0) got += (size_t)n;
    b->n_nl = (size_t)-1;
  } else {
    b->n_nl = (size_t)-1;
    /* exactly the reference's calls — gzread(fp, buf, 16384) under kseq (quack.c:152), default gzbuffer —
     * so that a damaged stream ends at the same byte: zlib drops the output of the call that fails */
    int n = 1;
    while (got + GZREAD_CHUNK <= BLOCK_BYTES && (n = gzread(s->gz, b->data + got, GZREAD_CHUNK)) > 0) got += (size_t)n;
  }
  return got;
}

static void *serial_main(void *arg) {
  qkh_source *s = arg;
  for (;;) {
    block *b = claim_block(s);
    size_t got;
    if (!b) break;
    got = fill_serial(s, b);
    if (!got) break;
    pthread_mutex_lock(&s->mu);
    b->len = got;
    b->ready = 1;
    s->head++;
    pthread_cond_broadcast(&s->filled);
    pthread_mutex_unlock(&s->mu);
  }
  finish_stream(s);
  return NULL;
}

/* the multi-threaded decoder handed its state over (qkh_pinflate_handoff): ring + serial producer from there */
static int start_serial_after_pgzip(qkh_source *s) {
  qkh_inflate *z = malloc(sizeof *z);
  uint8_t *hist = malloc(HIST);
  size_t hist_len = 0;
  if (qkh_pinflate_failed(s->pz)) (void)producer_oom(s);   /* (asked before the decoder is closed below) */
  if (!z || !hist) (void)producer_oom(s);
  if (!z || !hist || !qkh_pinflate_handoff(s->pz, z, hist, &hist_len)) {
    free(z);
    free(hist);
    return 0;
  }
  qkh_pinflate_close(s->pz);
  s->pz = NULL;
  s->zf = z;
  s->hist = hist;
  s->hist_len = hist_len;
  s->n_ring = SERIAL_RING;
  s->ring = calloc(s->n_ring, sizeof *s->ring);
  if (!s->ring) return (int)producer_oom(s);
  for (unsigned i = 0; i < s->n_ring; i++) {
    if (!(s->ring[i].base = malloc(HIST + BLOCK_BYTES))) return (int)producer_oom(s);
    s->ring[i].data = s->ring[i].base + HIST;
  }
  if (pthread_create(&s->threads[0], NULL, serial_main, s)) return 0;
  s->n_threads = 1;
  snprintf(s->kind, sizeof s->kind, "pgzip -> inflate_fast");
  return 1;
}

/* -------------------------------------------------------------------- bgzf */
/* total size of the BGZF member at p (0 if p does not start one) */
static size_t bgzf_member_size(const uint8_t *p, const uint8_t *end) {
  if (end - p < 28 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
  size_t xlen = p[10] | ((size_t)p[11] << 8);
  const uint8_t *x = p + 12, *xe = x + xlen;
  if (xe > end) return 0;
  while (xe - x >= 4) {
    size_t slen = x[2] | ((size_t)x[3] << 8);
    if (x[0] == 'B' && x[1] == 'C' && slen == 2 && xe - x >= 6) {
      size_t bsize = (x[4] | ((size_t)x[5] << 8)) + 1;
      return (bsize >= 12 + xlen + 8 && (size_t)(end - p) >= bsize) ? bsize : 0;
    }
    x += 4 + slen;
  }
  return 0;
}

static void *bgzf_dispatch_main(void *arg) {
  qkh_source *s = arg;
  const uint8_t *p = s->map, *end = s->map + s->map_len;
  while (p < end) {
    const uint8_t *start = p;
    size_t out = 0;
    while (p < end) {
      size_t bsize = bgzf_member_size(p, end), isize;
      if (!bsize) break;
      isize = p[bsize - 4] | ((size_t)p[bsize - 3] << 8) | ((size_t)p[bsize - 2] << 16) | ((size_t)p[bsize - 1] << 24);
      if (isize > 65536) break; /* not a BGZF block after all */
      if (out + isize > BLOCK_BYTES) break;
      out += isize;
      p += bsize;
    }
    if (p == start) break; /* a member that is not BGZF: the tail is decoded serially below */
    block *b = claim_block(s);
    if (!b) return NULL;
    pthread_mutex_lock(&s->mu);
    b->in = start;
    b->in_len = (size_t)(p - start);
    b->expect = out;
    s->head++;
    pthread_cond_signal(&s->work);
    pthread_mutex_unlock(&s->mu);
  }
  if (p < end) {
    /* rest of the file is ordinary gzip (or garbage): one serial decoder from
     * here; members are independent, so it needs no history */
    s->zf = malloc(sizeof *s->zf);
    s->hist = malloc(HIST);
    if (s->zf && s->hist) {
      qkh_inflate_init(s->zf, p, (size_t)(end - p));
      for (;;) {
        block *b = claim_block(s);
        size_t got;
        if (!b) return NULL;
        b->in = NULL;
        got = fill_serial(s, b);
        if (!got) break;
        pthread_mutex_lock(&s->mu);
        b->len = got;
        b->ready = 1;   /* in == NULL: the workers skip it */
        s->head++;
        pthread_cond_broadcast(&s->filled);
        pthread_mutex_unlock(&s->mu);
      }
    }
  }
  finish_stream(s);
  return NULL;
}

static void *plain_dispatch_main(void *arg) {
  qkh_source *s = arg;
  for (uint64_t off = 0; off < s->file_len; off += BLOCK_BYTES) {
    block *b = claim_block(s);
    if (!b) return NULL;
    pthread_mutex_lock(&s->mu);
    b->in = (const uint8_t *)s;   /* (a worker's block) */
    b->file_off = off;
    b->in_len = s->file_len - off < BLOCK_BYTES ? (size_t)(s->file_len - off) : (size_t)BLOCK_BYTES;
    b->expect = b->in_len;        /* a file that shrank meanwhile ends the stream there */
    s->head++;
    pthread_cond_signal(&s->work);
    pthread_mutex_unlock(&s->mu);
  }
  finish_stream(s);
  return NULL;
}

typedef struct {
  qkh_source *s;
  qkh_inflate *z;
} worker_arg;

static void *bgzf_worker_main(void *arg) {
  worker_arg *wa = arg;
  qkh_source *s = wa->s;
  qkh_inflate *z = wa->z;
  free(wa);
  for (;;) {
    block *b;
    size_t got = 0;
    long k;
    pthread_mutex_lock(&s->mu);
    for (;;) {
      /* blocks the dispatcher fills itself (ordinary gzip after the BGZF part) */
      while (s->next_work < s->head && !s->ring[s->next_work % s->n_ring].in) s->next_work++;
      if (s->next_work < s->head) break;
      if (s->stop || s->done) {
        pthread_mutex_unlock(&s->mu);
        return NULL;
      }
      pthread_cond_wait(&s->work, &s->mu);
    }
    b = &s->ring[s->next_work++ % s->n_ring];
    pthread_mutex_unlock(&s->mu);
    b->el.n = 0;
    if (s->plain_pool) {
      /* copying out of the page cache is what an uncompressed file costs (0.7 s of kernel time per 3.8 GB on
       * one thread): a few threads do it side by side */
      while (got < b->in_len && (k = (long)pread(s->fd, b->data + got, b->in_len - got, (off_t)(b->file_off + got))) > 0)
        got += (size_t)k;
    } else {
      qkh_inflate_init(z, b->in, b->in_len);
      int oom = 0;
      while (got < BLOCK_BYTES && !oom) {
        k = qkh_inflate_read(z, b->data + got, BLOCK_BYTES - got, got);
        oom = qkh_end_list_take(&b->el, z, got);
        if (k > 0) got += (size_t)k;
        else if (!(k == 0 && qkh_inflate_log_full(z))) break;
      }
      if (!oom) oom = qkh_end_list_crcs(&b->el, b->data, got);
      if (oom) {   /* the block's member ends / CRCs are incomplete: the stream ends in front of it, with an error */
        (void)producer_oom(s);
        got = 0;
        b->el.n = 0;
        b->expect = (size_t)-1;
      }
    }
    b->n_nl = qkh_index_lines(b->data, got, &b->nl, &b->cap_nl);
    pthread_mutex_lock(&s->mu);
    b->len = got;   /* != expect marks an undecodable run: the consumer ends the stream after it */
    b->ready = 1;
    pthread_cond_broadcast(&s->filled);
    pthread_mutex_unlock(&s->mu);
  }
}

/* --------------------------------------------------------------------- API */
/* GPUs of this node as the kernel driver lists them (KFD topology nodes with SIMDs), narrowed by
 * HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES — read from sysfs: asking the HIP runtime would start it
 * (0.15-0.3 s) on the thread that is about to parse. */
static int visible_gpus(void) {
  int n = 0;
  for (int node = 0; node < 64; node++) {
    char path[96], line[128];
    snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", node);
    FILE *f = fopen(path, "r");
    if (!f) break;
    while (fgets(line, sizeof line, f)) {
      long v;
      if (sscanf(line, "simd_count %ld", &v) == 1 && v > 0) n++;
    }
    fclose(f);
  }
  for (int k = 0; k < 2; k++) {
    const char *e = getenv(k ? "ROCR_VISIBLE_DEVICES" : "HIP_VISIBLE_DEVICES");
    if (e && *e) {
      int listed = 1;
      for (const char *c = e; *c; c++) listed += *c == ',';
      if (n == 0 || listed < n) n = listed;
    }
  }
  return n > 0 ? n : 1;
}

/* Decoder threads.  QUACK_THREADS, or this process's share of the host: the cores it may run on (affinity
 * mask, cgroup v2 CPU quota) divided by the node's GPUs — a node feeds one quack per GPU — and at most
 * MAX_WORKERS = 32 (round 2 capped at 16: the 1.8-Gbase file took 0.96-1.17 s with 16 threads, 0.50-0.58 s
 * with 32 on a 256-thread / 8-GPU host). */
static int n_cpus(void) {
  static int cached;
  const char *e = getenv("QUACK_THREADS");
  long n;
  if (e && atoi(e) > 0) {
    n = atoi(e);
  } else {
    if (cached) return cached;
    n = sysconf(_SC_NPROCESSORS_ONLN);
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
      long quota, period;
      if (fscanf(f, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0 && (quota + period - 1) / period < n)
        n = (quota + period - 1) / period;
      fclose(f);
    }
#endif
    n /= visible_gpus();
    if (n < 1) n = 1;
    if (n > MAX_WORKERS) n = MAX_WORKERS;
    cached = (int)n;
    return cached;
  }
  if (n < 1) n = 1;
  if (n > MAX_WORKERS) n = MAX_WORKERS;
  return (int)n;
}

/* Compressed bytes per slice of the multi-threaded inflate (pinflate.c).  Every slice costs the chain a hand-over — the last 32 KiB
 * resolved, the window copied, the next slice's thread woken — which nothing overlaps: on the GPU box's host config 2's file
 * (1.7 GB, 32 threads, tools/feed_bench) took 0.52 s at 512 KiB, 0.41 at 1 MiB, 0.40 at 2 MiB, 0.36 at 4 MiB and 8 MiB.  So: as
 * large as leaves every thread eight slices, between 1 and 4 MiB (a slot holds ~15 bytes per compressed byte: 60 MB at 4 MiB). */
static size_t pgzip_slice(size_t file_len, int threads) {
  const char *e = getenv("QUACK_PGZIP_CHUNK_KB");
  size_t s;
  if (e && atoi(e) > 0) return (size_t)atoi(e) << 10;
  s = file_len / ((size_t)(threads > 0 ? threads : 1) * 8u);
  s &= ~(((size_t)256 << 10) - 1u);
  if (s < ((size_t)1 << 20)) s = (size_t)1 << 20;
  if (s > ((size_t)4 << 20)) s = (size_t)4 << 20;
  /* ... within a memory budget: pinflate keeps threads + 4 slots of ~15 bytes per compressed byte — 36 slots of 4 MiB
   * slices are 2.2 GB of resident buffers.  The budget: 2.5 GB, a quarter of the cgroup's memory limit if that is less
   * (QUACK_PGZIP_MEM_MB overrides); slices shrink down to 1 MiB to stay inside it (2 MiB slices cost the feed ~10 %). */
  {
    const char *m = getenv("QUACK_PGZIP_MEM_MB");
    size_t budget = (size_t)2560 << 20;
    FILE *f = fopen("/sys/fs/cgroup/memory.max", "r");
    if (f) {
      unsigned long long lim = 0;
      if (fscanf(f, "%llu", &lim) == 1 && lim > 0 && lim / 4u < budget) budget = (size_t)(lim / 4u);   /* ("max": no limit) */
      fclose(f);
    }
    if (m && atoi(m) > 0) budget = (size_t)atoi(m) << 20;
    const size_t slots = (size_t)(threads > 0 ? threads : 1) + 4u;
    while (s > ((size_t)1 << 20) && s * 15u * slots > budget) s -= (size_t)256 << 10;
  }
  return s;
}

qkh_source *qkh_source_open(const char *path) {
  qkh_source *s = calloc(1, sizeof *s);
  struct stat st;
  int workers = 0;
  if (!s) return NULL;
  s->fd = -1;
  pthread_mutex_init(&s->mu, NULL);
  pthread_cond_init(&s->space, NULL);
  pthread_cond_init(&s->filled, NULL);
  pthread_cond_init(&s->work, NULL);
  /* gzopen like the reference (works on pipes too); regular files get the
   * faster producers */
  s->gz = gzopen(path, "rb");
  if (!s->gz) goto fail;
  snprintf(s->kind, sizeof s->kind, "zlib");
  if (stat(path, &st) == 0 && S_ISREG(st.st_mode)) {
    int fd = open(path, O_RDONLY);
    if (fd >= 0 && gzdirect(s->gz)) {
      gzclose(s->gz);
      s->gz = NULL;
      s->fd = fd;
      snprintf(s->kind, sizeof s->kind, "plain");
      if (n_cpus() > 1 && st.st_size >= 2 * (off_t)BLOCK_BYTES && !getenv("QUACK_NO_PLAIN_POOL")) {
        workers = n_cpus() < 4 ? n_cpus() : 4;
        s->plain_pool = 1;
        s->file_len = (uint64_t)st.st_size;
        snprintf(s->kind, sizeof s->kind, "plain x%d", workers);
      }
#ifdef POSIX_FADV_SEQUENTIAL
      posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
    } else if (fd >= 0 && st.st_size > 0 && !getenv("QUACK_ZLIB")) {
      void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      close(fd);
      if (m != MAP_FAILED) {
        madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        s->map = m;
        s->map_len = (size_t)st.st_size;
        gzclose(s->gz);
        s->gz = NULL;
        if (bgzf_member_size(s->map, s->map + s->map_len) && !getenv("QUACK_NO_BGZF")) {
          workers = n_cpus();
          snprintf(s->kind, sizeof s->kind, "bgzf x%d", workers);
        } else if (n_cpus() > 1 && !getenv("QUACK_NO_PGZIP") && s->map_len >= 2 * pgzip_slice(0, 1)) {
          s->pz = qkh_pinflate_open(s->map, s->map_len, n_cpus(), pgzip_slice(s->map_len, n_cpus()));
          if (!s->pz) goto fail;
          snprintf(s->kind, sizeof s->kind, "pgzip x%d", n_cpus());
          s->checks = 1;
          s->model.look = 1;   /* gzread starts by looking for a header */
          s->run_crc = (uint32_t)crc32(0L, Z_NULL, 0);
          return s;
        } else {
          s->zf = malloc(sizeof *s->zf);
          s->hist = malloc(HIST);
          if (!s->zf || !s->hist) goto fail;
          qkh_inflate_init(s->zf, s->map, s->map_len);
          snprintf(s->kind, sizeof s->kind, "inflate_fast");
        }
      }
    } else if (fd >= 0) {
      close(fd);
    }
  }
  s->checks = s->map != NULL;   /* bgzf, inflate_fast (pgzip returned above); zlib checks for itself */
  s->run_crc = (uint32_t)crc32(0L, Z_NULL, 0);
  s->model.look = 1;   /* gzread starts by looking for a header */
  s->n_ring = workers ? (unsigned)(2 * workers < SERIAL_RING ? SERIAL_RING : 2 * workers) : SERIAL_RING;
  s->ring = calloc(s->n_ring, sizeof *s->ring);
  if (!s->ring) goto fail;
  for (unsigned i = 0; i < s->n_ring; i++) {
    if (!(s->ring[i].base = malloc(HIST + BLOCK_BYTES))) goto fail;
    s->ring[i].data = s->ring[i].base + HIST;
  }
  if (workers) {
    if (pthread_create(&s->threads[s->n_threads], NULL, s->plain_pool ? plain_dispatch_main : bgzf_dispatch_main, s)) goto fail;
    s->n_threads++;
    for (int i = 0; i < workers; i++) {
      worker_arg *wa = malloc(sizeof *wa);
      s->worker_z[i] = malloc(sizeof(qkh_inflate));
      if (!wa || !s->worker_z[i]) {
        free(wa);
        break;
      }
      wa->s = s;
      wa->z = s->worker_z[i];
      s->n_workers = i + 1;
      if (pthread_create(&s->threads[s->n_threads], NULL, bgzf_worker_main, wa)) {
        free(wa);
        break;
      }
      s->n_threads++;
    }
    if (s->n_threads < 2) goto fail;
  } else {
    if (pthread_create(&s->threads[0], NULL, serial_main, s)) goto fail;
    s->n_threads = 1;
  }
  return s;
fail:
  qkh_source_close(s);
  return NULL;
}

/* the producer layer: next chunk in stream order, with its member ends */
static int raw_next(qkh_source *s, const uint8_t **data, size_t *len, const qkh_member_end **ends, unsigned *n_ends,
                    const uint32_t **piece_crc, const uint32_t **nl, size_t *n_nl) {
  *n_ends = 0;
  *n_nl = (size_t)-1;
  if (s->pz) {
    const int r = qkh_pinflate_next(s->pz, data, len);
    if (r) {
      qkh_pinflate_ends(s->pz, ends, n_ends, piece_crc);
      qkh_pinflate_lines(s->pz, nl, n_nl);
      return r;
    }
    /* the end — or a slice that outgrew the parallel decoder's memory bound: then the one-thread ring
     * producer takes over from the exact bit, in constant memory like the reference's gzread */
    if (!start_serial_after_pgzip(s)) return 0;
  }
  for (;;) {
    block *b;
    pthread_mutex_lock(&s->mu);
    if (s->holding) {
      s->holding = 0;
      s->tail++;
      pthread_cond_broadcast(&s->space);
    }
    while (!s->ended && !(s->tail < s->head && s->ring[s->tail % s->n_ring].ready) &&
           !(s->done && s->tail == s->head))
      pthread_cond_wait(&s->filled, &s->mu);
    if (s->ended || s->tail == s->head) {
      pthread_mutex_unlock(&s->mu);
      return 0;
    }
    b = &s->ring[s->tail % s->n_ring];
    if (b->in && b->len != b->expect) s->ended = 1;   /* deliver what it produced, then stop */
    *data = b->data;
    *len = b->len;
    *ends = b->el.ends;
    *n_ends = b->el.n;
    *piece_crc = b->el.piece_crc;
    *nl = b->nl;
    *n_nl = (s->checks || s->plain_pool) ? b->n_nl : (size_t)-1;
    s->holding = 1;
    pthread_mutex_unlock(&s->mu);
    if (*len || *n_ends) return 1;
    /* an empty run: take the next one */
  }
}

int qkh_source_next(qkh_source *s, const uint8_t **data, size_t *len) {
  const qkh_member_end *ends;
  const uint32_t *piece, *raw_nl;
  unsigned n_ends;
  size_t n_raw_nl;
  if (!s->checks) {
    /* plain files, zlib: the block as it is (with the line index of the plain pool's workers) */
    const int r = raw_next(s, data, len, &ends, &n_ends, &piece, &raw_nl, &n_raw_nl);
    s->plain_nl = (r && n_raw_nl != (size_t)-1) ? raw_nl : NULL;
    s->plain_n_nl = s->plain_nl ? n_raw_nl : 0;
    return r;
  }
  for (;;) {
    const uint8_t *raw;
    size_t raw_len, keep;
    if (s->crc_ended) return 0;
    /* the tail of the chunk that is about to be released has not been handed out yet */
    if (s->cur && s->n_held) memcpy(s->held, s->cur + s->cur_len - s->n_held, s->n_held);
    s->cur = NULL;
    if (!raw_next(s, &raw, &raw_len, &ends, &n_ends, &piece, &raw_nl, &n_raw_nl)) {
      /* the stream ended without a complaint: the held bytes are good */
      if (!s->n_held) return 0;
      *data = s->held;
      *len = s->n_held;
      s->n_held = 0;
      s->crc_ended = 1;   /* (nothing follows) */
      s->have_lines = 0;  /* (the tokenizer looks for the last few lines itself) */
      return 1;
    }
    /* the line index of front .. front + n_held + raw_len: the held bytes' lines out of the old index,
     * then the chunk's own */
    {
      const int o = s->cur_lines, n = o ^ 1;
      const size_t old_len = s->cur_len;   /* the old chunk's length; its last n_held bytes are the held ones */
      size_t cnt = 0;
      int ok = n_raw_nl != (size_t)-1 && (s->have_lines || s->delivered_raw == 0);
      if (ok) {
        size_t from = 0, keep_n = 0;
        if (s->n_held) {   /* first entry at or behind the held bytes' start (binary search) */
          size_t lo = 0, hi = s->n_lines[o];
          const size_t start = old_len - s->n_held;
          while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (s->lines[o][mid] < start) lo = mid + 1;
            else hi = mid;
          }
          from = lo;
          keep_n = s->n_lines[o] - lo;
        }
        if (keep_n + n_raw_nl > s->cap_lines[n]) {
          const size_t c = (keep_n + n_raw_nl) * 2 + 1024;
          uint32_t *g = realloc(s->lines[n], c * sizeof *g);
          if (g) {
            s->lines[n] = g;
            s->cap_lines[n] = c;
          } else {
            ok = 0;
          }
        }
        if (ok) {
          const uint32_t back = (uint32_t)(old_len - s->n_held);
          for (size_t i = 0; i < keep_n; i++) s->lines[n][i] = s->lines[o][from + i] - back;
          for (size_t i = 0; i < n_raw_nl; i++) s->lines[n][keep_n + i] = raw_nl[i] + (uint32_t)s->n_held;
          cnt = keep_n + n_raw_nl;
        }
      }
      s->cur_lines = n;
      s->n_lines[n] = cnt;
      s->have_lines = ok;
    }
    /* every chunk has >= 32 KiB of headroom in front of its data (ring blocks, pinflate slots) */
    uint8_t *front = (uint8_t *)raw - s->n_held;
    memcpy(front, s->held, s->n_held);
    /* chain the member CRCs through the chunk */
    size_t from = 0, good = raw_len;
    int bad = 0;
    for (unsigned i = 0; i <= n_ends && !bad; i++) {
      const size_t to = i < n_ends ? (ends[i].off < raw_len ? ends[i].off : raw_len) : raw_len;
      s->run_crc = (uint32_t)crc32_combine(s->run_crc, piece[i], (z_off_t)(to - from));
      s->run_len += to - from;
      if (i < n_ends) {
        uint64_t cut = 0;
        if (gzread_model_member(&s->model, s->delivered_raw + to, ends[i].bad_length || ends[i].crc != s->run_crc, &cut)) {
          /* the reference never sees the bytes from `cut` on (see gzread_model) */
          const uint64_t have_from = s->delivered_raw - s->n_held;   /* absolute offset of `front` */
          good = cut > have_from ? (size_t)(cut - have_from) : 0;
          bad = 1;
        }
        s->run_crc = (uint32_t)crc32(0L, Z_NULL, 0);
        s->run_len = 0;
      }
      from = to;
    }
    if (bad) {
      s->crc_ended = 1;
      s->n_held = 0;
      if (!good) return 0;
      *data = front;
      *len = good;
      s->given = good;
      return 1;
    }
    s->delivered_raw += raw_len;
    s->cur = front;
    s->cur_len = s->n_held + raw_len;
    keep = s->cur_len < HOLD_BACK ? s->cur_len : HOLD_BACK;
    *data = front;
    *len = s->cur_len - keep;
    s->n_held = keep;
    s->given = *len;
    if (*len) return 1;
    /* everything is held back: go on */
  }
}

int qkh_source_lines(qkh_source *s, const uint32_t **nl, size_t *n) {
  if (!s->checks) {
    if (!s->plain_nl) return 0;
    *nl = s->plain_nl;
    *n = s->plain_n_nl;
    return 1;
  }
  if (!s->have_lines) return 0;
  /* the entries inside the bytes handed out (the held-back tail has entries too) */
  const uint32_t *a = s->lines[s->cur_lines];
  size_t lo = 0, hi = s->n_lines[s->cur_lines];
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (a[mid] < s->given) lo = mid + 1;
    else hi = mid;
  }
  *nl = a;
  *n = lo;
  return 1;
}

const char *qkh_source_kind(const qkh_source *s) { return s->kind; }

int qkh_source_failed(const qkh_source *s) {
  return __atomic_load_n(&s->oom, __ATOMIC_RELAXED) || (s->pz && qkh_pinflate_failed(s->pz));
}

void qkh_source_close(qkh_source *s) {
  if (!s) return;
  pthread_mutex_lock(&s->mu);
  if (!s->stop) s->stop = 1;
  pthread_cond_broadcast(&s->space);
  pthread_cond_broadcast(&s->work);
  pthread_cond_broadcast(&s->filled);
  pthread_mutex_unlock(&s->mu);
  for (int i = 0; i < s->n_threads; i++) pthread_join(s->threads[i], NULL);
  if (s->pz && getenv("QUACK_VERBOSE")) {
    unsigned kept, redone;
    qkh_pinflate_stats(s->pz, &kept, &redone);
    fprintf(stderr, "[quack] %s: %u slices decoded speculatively, %u in order\n", s->kind, kept, redone);
  }
  qkh_pinflate_close(s->pz);
  pthread_mutex_destroy(&s->mu);
  pthread_cond_destroy(&s->space);
  pthread_cond_destroy(&s->filled);
  pthread_cond_destroy(&s->work);
  if (s->gz) gzclose(s->gz);
  if (s->fd >= 0) close(s->fd);
  if (s->map) munmap((void *)s->map, s->map_len);
  free(s->zf);
  free(s->hist);
  for (int i = 0; i < MAX_WORKERS; i++) free(s->worker_z[i]);
  if (s->ring)
    for (unsigned i = 0; i < s->n_ring; i++) {
      free(s->ring[i].base);
      free(s->ring[i].nl);
      qkh_end_list_free(&s->ring[i].el);
    }
  free(s->ring);
  free(s->lines[0]);
  free(s->lines[1]);
  free(s);
}
