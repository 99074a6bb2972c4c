/*
 * pinflate.c — one ordinary gzip stream, inflated by several threads.
 *
 * Most FASTQ arrives as plain `gzip` output: one DEFLATE stream whose every
 * block may refer to the 32 KiB before it, which is why zlib (and the
 * reference, quack.c:187-193) decode it on one core.  Here the compressed file
 * is cut into slices of equal compressed size and every slice is decoded
 * *speculatively* before its predecessor has finished:
 *
 *   find    the first position in the slice where a dynamic-Huffman block
 *           header parses (qkh_inflate_find_block);
 *   decode  from there into 16-bit elements: the 32 KiB in front of the slice's
 *           output are filled with markers 0x8000|i ("byte i of the unknown
 *           window"), so matches that reach into the unknown simply copy
 *           markers around (qkh_inflate_read16) — until the first block
 *           boundary at or past the next slice's nominal start;
 *   chain   strictly in slice order, and cheap: a slice's speculation is kept
 *           only if it began exactly where its predecessor's decoding ended
 *           (otherwise the start was not a real block boundary, or a stored /
 *           fixed block was skipped).  Then the predecessor's last 32 KiB
 *           resolve this slice's last 32 KiB, which releases the next slice;
 *   resolve markers -> bytes for the whole slice (a 64 KiB table lookup per
 *           element), in parallel with everybody else.
 *
 * A slice whose speculation cannot be kept — no header found, a decode error, a
 * start that does not line up, a member boundary closer than 32 KiB — is
 * decoded again, in order, by the ordinary byte decoder from the exact bit
 * where its predecessor stopped.  So the bytes delivered, and the byte at which
 * a damaged file stops, are those of the serial decoder by construction; the
 * speculation only ever saves time.  (The idea is that of pugz / rapidgzip;
 * the code is not.)
 */
#include "pinflate.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include "inflate_fast.h"

enum { WIN = 32768, MAX_THREADS = 32 };
/* a spin-wait's breather (the x86 builtin alone kept the host library from building anywhere else) */
static inline void cpu_relax(void) {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ volatile("yield");
#else
  __asm__ volatile("" ::: "memory");
#endif
}
/* memory bounds: a slice that inflates past SPEC_MAX bytes is not worth speculating on (FASTQ is 4-5x;
 * this is 32x) and is decoded in order instead; buffers that grew past KEEP_MAX are given back after use */
#define SPEC_MAX ((size_t)32 << 20)
#define KEEP_MAX ((size_t)64 << 20)
/* ... and the in-order decoder, which has to take whatever a slice inflates to (1 MiB of zeros: 1 GiB),
 * stops at INORDER_MAX bytes and hands its state over: the caller goes on with the one-thread ring
 * producer, which streams in constant memory like the reference's gzread (qkh_pinflate_handoff) */
#define INORDER_MAX ((size_t)256 << 20)

typedef struct {
  uint16_t *b16;   /* WIN markers + speculative output */
  size_t cap16;    /* output elements that fit */
  uint8_t *b8;     /* WIN bytes of history + output */
  size_t cap8;
  size_t len;      /* output bytes of the slice, at b8 + WIN */
  int ready;
  /* member trailers inside the slice and the CRC-32 of the pieces they cut it into, computed by
   * the worker once the bytes are final */
  qkh_end_list el;
  uint32_t *nl;    /* offsets of the newlines in the slice (qkh_index_lines), by the worker */
  size_t n_nl, cap_nl;
} pslot;

struct qkh_pinflate {
  const uint8_t *in;
  size_t in_len, slice;
  unsigned n_slices, n_slots;
  pslot *slots;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  unsigned next_slice;   /* next slice without a worker */
  unsigned tail;         /* slices consumed */
  int holding, stop, failed;
  /* the chain: state of the stream in front of slice `chain_next` */
  _Atomic unsigned chain_next;   /* (written under mu; the slice whose turn is next also watches it without) */
  uint64_t chain_bit;
  uint8_t chain_win[WIN];
  size_t chain_win_len, chain_member_out;
  unsigned chain_members;
  int chain_end;         /* the stream ended (or broke) inside an earlier slice */
  int n_threads;
  pthread_t threads[MAX_THREADS];
  /* a slice outgrew the in-order decoder's bound: its decoder, frozen in mid-stream, and the window */
  int handoff;
  qkh_inflate handoff_z;
  size_t inorder_max;
  /* statistics (tests, QUACK_VERBOSE) */
  unsigned kept, redone;
};

static int grow16(pslot *s, size_t need) {
  if (need <= s->cap16) return 0;
  size_t cap = s->cap16 + s->cap16 / 2;
  if (cap < need) cap = need;
  uint16_t *nb = realloc(s->b16, (WIN + cap + 64) * sizeof *nb);
  if (!nb) return -1;
  if (!s->b16)
    for (unsigned i = 0; i < WIN; i++) nb[i] = (uint16_t)(0x8000u | i);
  s->b16 = nb;
  s->cap16 = cap;
  return 0;
}

static int grow8(pslot *s, size_t need) {
  if (need <= s->cap8) return 0;
  size_t cap = s->cap8 + s->cap8 / 2;
  if (cap < need) cap = need;
  uint8_t *nb = realloc(s->b8, WIN + cap + 64);
  if (!nb) return -1;
  s->b8 = nb;
  s->cap8 = cap;
  return 0;
}

/* bit where slice k nominally begins */
static uint64_t slice_bit(const qkh_pinflate *p, unsigned k) { return (uint64_t)k * p->slice * 8u; }

typedef struct {
  int found, ok, end;
  uint64_t start_bit, end_bit;
  size_t n;
} spec_result;

static void speculate(qkh_pinflate *p, unsigned k, pslot *s, qkh_inflate *z, spec_result *r) {
  memset(r, 0, sizeof *r);
  const uint64_t to = k + 1 < p->n_slices ? slice_bit(p, k + 1) : (uint64_t)p->in_len * 8u;
  const int64_t at = qkh_inflate_find_block(p->in, p->in_len, slice_bit(p, k), to, z);
  if (at < 0) return;
  r->found = 1;
  r->start_bit = (uint64_t)at;
  qkh_inflate_init_at(z, p->in, p->in_len, (uint64_t)at, 0, 0, 1);
  z->stop_bit = k + 1 < p->n_slices ? slice_bit(p, k + 1) : 0;
  if (grow16(s, p->slice * 5)) return;
  size_t n = 0;
  for (;;) {
    if (n == s->cap16 && grow16(s, n + 1)) return;
    long got = qkh_inflate_read16(z, s->b16 + WIN + n, s->cap16 - n, WIN + n);
    if (qkh_end_list_take(&s->el, z, n)) return;
    if (got > 0) n += (size_t)got;
    if (n > SPEC_MAX) return;              /* see SPEC_MAX */
    if (z->state == QKH_Z_ERROR) return;   /* garbage, or a damaged file: decided in order */
    if (z->stopped || z->state == QKH_Z_DONE) break;
    if (got <= 0) return;
  }
  r->ok = 1;
  r->end = z->state == QKH_Z_DONE;
  r->end_bit = qkh_inflate_bitpos(z);
  r->n = n;
}

/* markers -> bytes: dst[i] = lut[src[i]] (lut: identity below 256, the previous window from 0x8000).
 * In FASTQ the markers are the few bytes of a record that were copied out of the unknown window (and copies of those copies:
 * the `@r...` of every header line) — a run of 32 elements is free of them nine times out of ten, and such a run is narrowed
 * by two vector instructions instead of 32 table look-ups.  Measured on one thread (tools/inflate_bench, "worker stages"):
 * 1.9 GB/s -> see profiles/r05_inflate_stages.log; the pass was a seventh of what a decoder thread does with a slice. */
static void resolve_scalar(const uint16_t *src, uint8_t *dst, size_t n, const uint8_t *lut) {
  size_t i = 0;
  for (; i + 4 <= n; i += 4) {
    dst[i] = lut[src[i]];
    dst[i + 1] = lut[src[i + 1]];
    dst[i + 2] = lut[src[i + 2]];
    dst[i + 3] = lut[src[i + 3]];
  }
  for (; i < n; i++) dst[i] = lut[src[i]];
}
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
__attribute__((target("avx2"))) static void resolve_avx2(const uint16_t *src, uint8_t *dst, size_t n, const uint8_t *lut) {
  const __m256i high = _mm256_set1_epi16((short)0xFF00);
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i));
    const __m256i b = _mm256_loadu_si256((const __m256i *)(src + i + 16));
    if (_mm256_testz_si256(_mm256_or_si256(a, b), high)) {   /* 32 plain bytes */
      const __m256i pk = _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8);   /* (packus works per 128-bit lane) */
      _mm256_storeu_si256((__m256i *)(dst + i), pk);
    } else {
      for (size_t j = i; j < i + 32; j++) dst[j] = lut[src[j]];
    }
  }
  for (; i < n; i++) dst[i] = lut[src[i]];
}
#endif
void qkh_resolve16(const uint16_t *src, uint8_t *dst, size_t n, const uint8_t *lut) {
#if defined(__x86_64__) && defined(__GNUC__)
  static int have = -1;   /* (benign race: every thread computes the same answer) */
  if (have < 0) have = __builtin_cpu_supports("avx2") ? 1 : 0;
  if (have) {
    resolve_avx2(src, dst, n, lut);
    return;
  }
#endif
  resolve_scalar(src, dst, n, lut);
}
#define resolve qkh_resolve16

typedef struct {
  qkh_pinflate *p;
  qkh_inflate *z;
  uint8_t *lut;   /* 65536: identity below 256, the previous window from 0x8000 */
} worker;

static void *worker_main(void *arg) {
  worker *w = arg;
  qkh_pinflate *p = w->p;
  qkh_inflate *z = w->z;
  uint8_t *lut = w->lut;
  for (unsigned i = 0; i < 256; i++) lut[i] = (uint8_t)i;
  for (;;) {
    unsigned k;
    pslot *s;
    spec_result sp;
    pthread_mutex_lock(&p->mu);
    while (!p->stop && !p->chain_end && p->next_slice < p->n_slices && p->next_slice - p->tail >= p->n_slots)
      pthread_cond_wait(&p->cv, &p->mu);
    if (p->stop || p->chain_end || p->next_slice >= p->n_slices) {
      pthread_mutex_unlock(&p->mu);
      break;
    }
    k = p->next_slice++;
    s = &p->slots[k % p->n_slots];
    s->ready = 0;
    s->len = 0;
    s->el.n = 0;
    pthread_mutex_unlock(&p->mu);

    memset(&sp, 0, sizeof sp);
    if (k > 0) speculate(p, k, s, z, &sp);

    /* ---- my turn in the chain.  The slice that is next (or next but one) does not sleep: waking a thread through the condition
     * variable took longer than the hand-over itself (resolving 32 KiB), and the chain is the one thing nothing overlaps */
    for (int spins = 0; spins < 20000; spins++) {
      const unsigned cn = atomic_load_explicit(&p->chain_next, memory_order_acquire);
      if (cn == k || k - cn > 2u) break;
      cpu_relax();
    }
    pthread_mutex_lock(&p->mu);
    while (!p->stop && p->chain_next != k) pthread_cond_wait(&p->cv, &p->mu);
    if (p->stop) {
      pthread_mutex_unlock(&p->mu);
      break;
    }
    /* chain_* belong to this thread until it advances chain_next */
    const int ended = p->chain_end;
    pthread_mutex_unlock(&p->mu);

    size_t n = 0, member_out = 0;
    unsigned members = 0;
    uint64_t end_bit = 0;
    int end = ended, failed = 0, keep = 0, overflow = 0;
    uint8_t new_win[WIN];
    size_t new_win_len = 0;

    if (!ended) {
      keep = k > 0 && sp.found && sp.ok && sp.start_bit == p->chain_bit && p->chain_member_out >= WIN &&
             (!z->pend_set || (uint32_t)(p->chain_member_out + z->pend_out) == z->pend_isize) &&
             grow8(s, sp.n) == 0;
      if (keep) {
        n = sp.n;
        memcpy(lut + 0x8000, p->chain_win, WIN);
        if (n >= WIN) {
          resolve(s->b16 + WIN + n - WIN, new_win, WIN, lut);
        } else {
          memcpy(new_win, p->chain_win + n, WIN - n);
          resolve(s->b16 + WIN, new_win + WIN - n, n, lut);
        }
        new_win_len = WIN;
        member_out = z->base_unknown ? p->chain_member_out + n : z->member_out;
        members = p->chain_members + z->members;
        end_bit = sp.end_bit;
        end = sp.end;
      } else {
        /* in order, from the exact bit the previous slice stopped at */
        s->el.n = 0;
        if (k == 0) qkh_inflate_init(z, p->in, p->in_len);
        else qkh_inflate_init_at(z, p->in, p->in_len, p->chain_bit, p->chain_member_out, p->chain_members, 0);
        z->stop_bit = k + 1 < p->n_slices ? slice_bit(p, k + 1) : 0;
        if (grow8(s, p->slice * 5)) failed = 1;
        else memcpy(s->b8 + WIN - p->chain_win_len, p->chain_win, p->chain_win_len);
        while (!failed) {
          if (n == s->cap8 && grow8(s, n + 1)) {
            failed = 1;
            break;
          }
          long got = qkh_inflate_read(z, s->b8 + WIN + n, s->cap8 - n, p->chain_win_len + n);
          if (qkh_end_list_take(&s->el, z, n)) failed = 1;
          if (got > 0) n += (size_t)got;
          if (z->stopped || z->state == QKH_Z_DONE || z->state == QKH_Z_ERROR) break;
          if (got <= 0 && !(got == 0 && qkh_inflate_log_full(z))) break;   /* (a full trailer log is not the end) */
          if (n >= p->inorder_max) {   /* enough for one slot: the rest of the stream goes to the serial producer */
            overflow = 1;
            break;
          }
        }
        end = failed || overflow || !z->stopped;   /* finished, handed over, or broken at this byte: nothing follows */
        end_bit = qkh_inflate_bitpos(z);
        member_out = z->member_out;
        members = z->members;
        if (!failed) {
          new_win_len = p->chain_win_len + n < WIN ? p->chain_win_len + n : WIN;
          memcpy(new_win, s->b8 + WIN + n - new_win_len, new_win_len);
        }
        if (!failed && qkh_end_list_crcs(&s->el, s->b8 + WIN, n)) failed = 1;
        if (!failed) s->n_nl = qkh_index_lines(s->b8 + WIN, n, &s->nl, &s->cap_nl);
      }
    }

    pthread_mutex_lock(&p->mu);
    if (!ended) {
      memcpy(p->chain_win, new_win, new_win_len);
      p->chain_win_len = new_win_len;
      p->chain_bit = end_bit;
      p->chain_member_out = member_out;
      p->chain_members = members;
      p->chain_end = end;
      if (failed) p->failed = 1;
      if (overflow && !failed) {
        qkh_inflate_clone(&p->handoff_z, z);
        p->handoff_z.stop_bit = 0;
        p->handoff_z.stopped = 0;
        p->handoff = 1;
      }
      if (keep) p->kept++;
      else p->redone++;
    }
    p->chain_next = k + 1;
    if (!keep) {
      s->len = failed ? 0 : n;
      s->ready = 1;
    }
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);

    if (keep) {
      /* the bulk of the work, off the chain: lut still holds the previous window */
      resolve(s->b16 + WIN, s->b8 + WIN, n, lut);
      const int crc_failed = qkh_end_list_crcs(&s->el, s->b8 + WIN, n);
      s->n_nl = qkh_index_lines(s->b8 + WIN, n, &s->nl, &s->cap_nl);
      pthread_mutex_lock(&p->mu);
      if (crc_failed) p->failed = 1;
      s->len = crc_failed ? 0 : n;
      s->ready = 1;
      pthread_cond_broadcast(&p->cv);
      pthread_mutex_unlock(&p->mu);
    }
  }
  free(w->z);
  free(w->lut);
  free(w);
  return NULL;
}

qkh_pinflate *qkh_pinflate_open(const uint8_t *data, size_t len, int threads, size_t slice_bytes) {
  qkh_pinflate *p = calloc(1, sizeof *p);
  if (!p) return NULL;
  if (threads < 1) threads = 1;
  if (threads > MAX_THREADS) threads = MAX_THREADS;
  if (slice_bytes < 4096) slice_bytes = 4096;
  p->in = data;
  p->in_len = len;
  p->slice = slice_bytes;
  p->n_slices = (unsigned)((len + slice_bytes - 1) / slice_bytes);
  if (!p->n_slices) p->n_slices = 1;
  p->n_slots = (unsigned)threads + 4;
  {
    const char *e = getenv("QUACK_PGZIP_MAX_SLICE_MB");   /* (tests) */
    p->inorder_max = e && atoi(e) > 0 ? (size_t)atoi(e) << 20 : INORDER_MAX;
  }
  p->slots = calloc(p->n_slots, sizeof *p->slots);
  pthread_mutex_init(&p->mu, NULL);
  pthread_cond_init(&p->cv, NULL);
  if (!p->slots) goto fail;
  for (int i = 0; i < threads; i++) {
    worker *w = calloc(1, sizeof *w);
    if (w) {
      w->p = p;
      w->z = malloc(sizeof *w->z);
      w->lut = calloc(1, 65536);
    }
    if (!w || !w->z || !w->lut || pthread_create(&p->threads[p->n_threads], NULL, worker_main, w)) {
      if (w) {
        free(w->z);
        free(w->lut);
      }
      free(w);
      break;
    }
    p->n_threads++;
  }
  if (!p->n_threads) goto fail;
  return p;
fail:
  qkh_pinflate_close(p);
  return NULL;
}

int qkh_pinflate_next(qkh_pinflate *p, const uint8_t **data, size_t *len) {
  for (;;) {
    pslot *s;
    pthread_mutex_lock(&p->mu);
    if (p->holding) {
      pslot *h = &p->slots[p->tail % p->n_slots];
      if (h->cap8 > KEEP_MAX || h->cap16 > KEEP_MAX) {   /* an unusually compressible slice: do not keep its buffers */
        free(h->b8);
        free(h->b16);
        h->b8 = NULL;
        h->b16 = NULL;
        h->cap8 = h->cap16 = 0;
      }
      p->holding = 0;
      p->tail++;
      pthread_cond_broadcast(&p->cv);
    }
    for (;;) {
      if (p->tail < p->next_slice && p->slots[p->tail % p->n_slots].ready) break;
      if (p->tail == p->n_slices || (p->chain_end && p->tail == p->next_slice) || p->stop) {
        pthread_mutex_unlock(&p->mu);
        return 0;
      }
      pthread_cond_wait(&p->cv, &p->mu);
    }
    s = &p->slots[p->tail % p->n_slots];
    *data = s->b8 + WIN;
    *len = s->len;
    p->holding = 1;
    pthread_mutex_unlock(&p->mu);
    if (*len || s->el.n) return 1;   /* (a slice may hold nothing but a member's trailer) */
  }
}

void qkh_pinflate_ends(qkh_pinflate *p, const qkh_member_end **ends, unsigned *n_ends, const uint32_t **piece_crc) {
  const pslot *s = &p->slots[p->tail % p->n_slots];   /* the slice handed out by the last qkh_pinflate_next */
  *ends = s->el.ends;
  *n_ends = p->holding ? s->el.n : 0;
  *piece_crc = s->el.piece_crc;
}

void qkh_pinflate_lines(qkh_pinflate *p, const uint32_t **nl, size_t *n) {
  const pslot *s = &p->slots[p->tail % p->n_slots];
  *nl = s->nl;
  *n = p->holding ? s->n_nl : (size_t)-1;
}

int qkh_pinflate_handoff(qkh_pinflate *p, qkh_inflate *z, uint8_t *window, size_t *window_len) {
  int h;
  pthread_mutex_lock(&p->mu);
  h = p->handoff && (p->chain_end && p->tail == p->next_slice);   /* ... and everything before it was taken */
  if (h) {
    qkh_inflate_clone(z, &p->handoff_z);
    memcpy(window, p->chain_win, p->chain_win_len);
    *window_len = p->chain_win_len;
  }
  pthread_mutex_unlock(&p->mu);
  return h;
}

int qkh_pinflate_failed(qkh_pinflate *p) {
  int f;
  pthread_mutex_lock(&p->mu);
  f = p->failed;
  pthread_mutex_unlock(&p->mu);
  return f;
}

void qkh_pinflate_stats(const qkh_pinflate *p, unsigned *kept, unsigned *redone) {
  *kept = p->kept;
  *redone = p->redone;
}

void qkh_pinflate_close(qkh_pinflate *p) {
  if (!p) return;
  pthread_mutex_lock(&p->mu);
  p->stop = 1;
  pthread_cond_broadcast(&p->cv);
  pthread_mutex_unlock(&p->mu);
  for (int i = 0; i < p->n_threads; i++) pthread_join(p->threads[i], NULL);
  pthread_mutex_destroy(&p->mu);
  pthread_cond_destroy(&p->cv);
  if (p->slots)
    for (unsigned i = 0; i < p->n_slots; i++) {
      free(p->slots[i].b16);
      free(p->slots[i].b8);
      qkh_end_list_free(&p->slots[i].el);
      free(p->slots[i].nl);
    }
  free(p->slots);
  free(p);
}
