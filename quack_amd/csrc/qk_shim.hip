// qk_shim.hip — implementation of include/quack_hip.h (libquack_hip.so).
//
// One qk_accum == the state of one read_fastq() call (quack.c:180-228) kept
// resident on one MI355X: a planar u64 counter table [97][table_len], two
// pinned host batch slots with matching device slots (hipHostMalloc +
// hipMemcpyAsync, one HIP stream per slot so the copy of batch k+1 overlaps
// the kernels of batch k), and the adapter 10-mer tables.
// No CPU fallback lives here: every failure is reported to the caller.
#include "quack_hip.h"

#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <utility>
#include <vector>

#include "qk_adapter_kernels.hip.h"
#include "qk_kernels.hip.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define QK_HIP(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess)                                                   \
      return fail(QK_EHIP, "%s failed: %s (%s:%d)", #call,                  \
                  hipGetErrorString(e_), __FILE__, __LINE__);               \
  } while (0)

uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// QUACK_VERBOSE: where the start-up of an accumulator goes (stderr), the question behind every end-to-end number
struct Lap {
  const bool on = getenv("QUACK_VERBOSE") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  char line[512] = "";
  void mark(const char *what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    const size_t l = strlen(line);
    snprintf(line + l, sizeof line - l, " %s %.1f ms,", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
  void print(const char *head) {
    if (on) fprintf(stderr, "[quack] %s:%s\n", head, line);
  }
};

int env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

// Experiment switches (round 5): the product library reads no environment variable that selects another launch geometry or
// kernel variant — only QUACK_VERBOSE, QUACK_HIP_BATCH_MB, QUACK_HIP_CHECK_PADS and the QUACK_HIP_NO_* / _UNFUSED_ADAPTERS
// fallbacks.  Everything the tools and the parity tests use to force a geometry the planner would not pick lives in ONE variable,
//     QUACK_HIP_TUNE="key=value,key,..."      (group=2, small_ring, pipe=3, threads=512, ...)
// which only a -DQK_EXPERIMENT build parses (quack_amd/libquack_hip_exp.so, tools/kbench); that build also carries the kernel
// variants those geometries need.  In the product build every switch is a compile-time constant.
#ifdef QK_EXPERIMENT
const char *tune_find(const char *key) {   // -> the text behind "key=" ("" for a bare key), or nullptr
  const char *s = getenv("QUACK_HIP_TUNE");
  const size_t kl = strlen(key);
  while (s && *s) {
    const char *e = strchr(s, ',');
    const size_t len = e ? (size_t)(e - s) : strlen(s);
    if (len >= kl && strncmp(s, key, kl) == 0 && (len == kl || s[kl] == '=')) return len == kl ? "" : s + kl + 1;
    s = e ? e + 1 : nullptr;
  }
  return nullptr;
}
bool tune_on(const char *key) { return tune_find(key) != nullptr; }
int tune_int(const char *key, int dflt) {
  const char *v = tune_find(key);
  return (v && *v) ? atoi(v) : (v ? 1 : dflt);
}
#else
inline bool tune_on(const char *) { return false; }
inline int tune_int(const char *, int dflt) { return dflt; }
#endif

struct Slot {
  uint8_t *h_seq = nullptr, *h_qual = nullptr;
  uint64_t *h_off = nullptr;
  uint32_t *h_len = nullptr;          // gapped batches only
  uint8_t *d_seq = nullptr, *d_qual = nullptr;
  uint64_t *d_off = nullptr;
  uint32_t *d_len = nullptr;
  uint32_t *d_hit = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  bool busy = false;
};

struct TimedLaunch {
  hipEvent_t t0, t1;   // around the histogram kernel (the dominant kernel)
  hipEvent_t b0, b1;   // around every kernel of the batch (pre-pass ... adapter count); may alias t0 / t1
};

}  // namespace

constexpr int kSets = 2;   // order / reach buffer sets: the pre-pass of batch k + 1 runs while the histogram kernel of batch k reads its own
                           // (a third set, so that no pre-pass starts together with a histogram kernel: measured, no difference)

struct qk_accum {
  int device = 0;
  int n_cu = 256;
  // counter table: planar [QK_N_ROWS][table_len] u64 + 1 trailing word
  unsigned long long *d_table = nullptr;
  uint32_t *d_table32 = nullptr;      // what the histogram kernels flush into (qk::HistParams::table32); folded into d_table by settle()
  uint64_t reads32 = 0;               // reads that may have been counted into d_table32 since it was last folded
  uint64_t table_len = 0;
  uint64_t max_len = 0;
  uint64_t n_reads = 0;
  // adapters
  bool adapters = false;
  uint32_t *d_kmer_bits = nullptr;
  uint32_t *d_kmer_filter = nullptr;
  uint32_t filter_bits = 0;
  uint4 *d_kmer_buckets = nullptr;    // exact LDS table of the fused path (may be absent)
  uint32_t bucket_log2 = 0, bucket_mul = 0;
  // pipeline
  Slot slot[2];
  int next_slot = 0;
  int held_slot = -1;
  uint64_t cap_bytes = 0, cap_reads = 0;
  hipStream_t stream = nullptr;     // own stream for device-resident submits
  bool foreign_stream_used = false;
  uint32_t *d_queues = nullptr;       // ring of work-queue counters (multi-tile launches)
  unsigned queue_seq = 0;
  uint32_t *d_hit_scratch = nullptr;  // first-hit buffer for device submits
  uint64_t hit_scratch_reads = 0;
  unsigned long long *d_starts_scratch = nullptr;   // strided batches restated as gapped ones (see enqueue_batch)
  uint64_t starts_scratch_reads = 0;
  // launches of one accumulator run in submission order even when they come
  // from different streams: they share the queue ring, the first-hit scratch
  // and the table's flush targets
  // long ragged reads: reads ordered by the tiles they reach (qk::reach_* kernels)
  // (two sets, used in turn: the pre-pass of batch k+1 may run — on the side stream below — while the histogram
  // kernel of batch k still reads its own)
  uint32_t *d_order[kSets] = {};       // [order_cap]
  uint64_t order_cap[kSets] = {};
  uint32_t *d_reach[kSets] = {};       // [kReachMaxTiles] reach | [kReachMaxTiles + 1] counts | [kReachMaxTiles + 1] cursor
  unsigned set_turn = 0;
  // Side stream (round 3): the reach sort of long ragged reads, which only reads the batch's lengths, runs here when
  // the batch was submitted on the accumulator's OWN stream
  // (device-resident feed: the caller's buffers are complete when the call is made, that stream is ordered with
  // nothing of the caller's).  It then overlaps the tail of the previous batch's histogram kernel (its flush, its
  // teardown, in which the CUs drain) instead of standing between two kernels — as far as the CUs have room: the
  // histogram kernel holds every VGPR of a CU until its workgroup retires.  Measured on 1-20 kb reads, four alternations
  // of 200 steps on one box: 0.6128 -> 0.6100 ms per step (3-4 of the 30 us around a 0.58 ms kernel).  Batches from the pinned slots keep everything on the slot's
  // stream (their copies come first, and the two slots overlap each other anyway).
  hipStream_t side = nullptr;
  hipEvent_t side_done = nullptr;                   // behind a pre-pass on `side`: the histogram kernel waits for it
  hipEvent_t set_free[kSets] = {};     // behind the histogram kernel that read set i
  bool set_busy[kSets] = {};
  unsigned pads_checked = 0;          // neutral-pad batches whose pads were verified on the device (the first two always are)
  uint32_t *d_status = nullptr;       // device word: bit 0 = an "aligned" batch was not aligned
  bool status_armed = false;
  hipEvent_t order_ev = nullptr;
  hipStream_t order_stream = nullptr;
  bool order_valid = false;
  bool order_recorded = false;        // order_ev already stands behind the last launch on order_stream
  // tuning
  int threads = 1024, unroll = 0, pipe = 0, tile = 0, wgs_per_cu = 0;   // 0 = automatic
  // timing
  int timing = 0;                   // 0 off; N: events around every Nth batch
  uint64_t timing_seq = 0;
  std::vector<TimedLaunch> timed;
  std::vector<hipEvent_t> event_pool;
  double timing_ms = 0, timing_batch_ms = 0, timing_min_ms = 0, timing_max_ms = 0;
  uint64_t timing_launches = 0;
};

namespace {

// The pinned slot `i` and its device twin, allocated on first use: page-locking 150 MB takes
// ~40 ms, and the second slot's share of that overlaps the tokenizer filling the first.
int ensure_slot(qk_accum *a, int i) {
  Slot &s = a->slot[i];
  if (s.h_seq) return QK_OK;
  if (!a->cap_bytes) {
    const uint64_t mb = (uint64_t)env_int("QUACK_HIP_BATCH_MB", 32);
    a->cap_bytes = mb << 20;
    if (const int kb = env_int("QUACK_HIP_BATCH_KB", 0)) a->cap_bytes = (uint64_t)kb << 10;   // (tests: tiny slots)
    a->cap_reads = a->cap_bytes / 32 + 1024;  // >= one read per 32 bytes
  }
  Lap lap;
  QK_HIP(hipHostMalloc((void **)&s.h_seq, a->cap_bytes + QK_TAIL_SLACK, hipHostMallocDefault));
  QK_HIP(hipHostMalloc((void **)&s.h_qual, a->cap_bytes + QK_TAIL_SLACK, hipHostMallocDefault));
  QK_HIP(hipHostMalloc((void **)&s.h_off, (a->cap_reads + 1) * sizeof(uint64_t), hipHostMallocDefault));
  QK_HIP(hipHostMalloc((void **)&s.h_len, a->cap_reads * sizeof(uint32_t), hipHostMallocDefault));
  lap.mark("pinned");
  QK_HIP(hipMalloc((void **)&s.d_len, a->cap_reads * sizeof(uint32_t)));
  QK_HIP(hipMalloc((void **)&s.d_seq, a->cap_bytes + QK_TAIL_SLACK));
  QK_HIP(hipMalloc((void **)&s.d_qual, a->cap_bytes + QK_TAIL_SLACK));
  QK_HIP(hipMalloc((void **)&s.d_off, (a->cap_reads + 1) * sizeof(uint64_t)));
  if (a->adapters) QK_HIP(hipMalloc((void **)&s.d_hit, a->cap_reads * sizeof(uint32_t)));
  QK_HIP(hipMemset(s.d_seq + a->cap_bytes, 0, QK_TAIL_SLACK));
  QK_HIP(hipMemset(s.d_qual + a->cap_bytes, 0, QK_TAIL_SLACK));
  lap.mark("device");
  QK_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
  QK_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  lap.mark("stream");
  lap.print(i ? "slot 1" : "slot 0");
  return QK_OK;
}

// Grow the planar table so that it holds `need` positions.  Rare (once or
// twice per file, like the realloc at quack.c:194-198), so it synchronises.
int fold_table32(qk_accum *a, hipStream_t st) {
  if (!a->d_table32 || a->reads32 == 0) return QK_OK;
  const size_t n = (size_t)QK_N_ROWS * a->table_len;
  hipLaunchKernelGGL(qk::table_fold_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, a->d_table,
                     a->d_table32, n);
  QK_HIP(hipGetLastError());
  a->reads32 = 0;
  return QK_OK;
}

int grow_table(qk_accum *a, uint64_t need, bool exact = false) {
  if (need <= a->table_len) return QK_OK;
  QK_HIP(hipDeviceSynchronize());
  {   // the 32-bit side table goes into the old table first; the new one starts empty
    int rc = fold_table32(a, a->stream);
    if (rc) return rc;
    QK_HIP(hipStreamSynchronize(a->stream));
    if (a->d_table32) QK_HIP(hipFree(a->d_table32));
    a->d_table32 = nullptr;
  }
  // amortised doubling while reads arrive; `exact` when several accumulators
  // must agree on one geometry before their tables are summed
  uint64_t nl = exact ? need : std::max<uint64_t>(need, a->table_len * 2);
  nl = round_up(std::max<uint64_t>(nl, 64), 64);
  unsigned long long *nt = nullptr;
  const size_t words = (size_t)QK_N_ROWS * nl + 1;
  QK_HIP(hipMalloc((void **)&nt, words * sizeof(unsigned long long)));
  QK_HIP(hipMemset(nt, 0, words * sizeof(unsigned long long)));
  if (a->d_table) {
    QK_HIP(hipMemcpy2D(nt, nl * 8, a->d_table, a->table_len * 8,
                       a->table_len * 8, QK_N_ROWS, hipMemcpyDeviceToDevice));
    QK_HIP(hipFree(a->d_table));
  }
  a->d_table = nt;
  a->table_len = nl;
  QK_HIP(hipMalloc((void **)&a->d_table32, (size_t)QK_N_ROWS * nl * sizeof(uint32_t)));
  QK_HIP(hipMemset(a->d_table32, 0, (size_t)QK_N_ROWS * nl * sizeof(uint32_t)));
  return QK_OK;
}

struct Plan {
  uint32_t n_tiles, tile_pos, ch, rw;
  int unroll, pipe;
  uint32_t stage_reads;
  uint64_t reads_per_slice, n_slices, n_blocks;
  uint32_t bucket_log2, halo, replicas;
  bool fused_adapters, dynamic, aligned, sorted;
  bool w16;   // 16 positions per lane (qk::hist_kernel<..., W16>): dword-aligned batches under the planner's own geometry
  uint32_t fh_words;   // first-hit ring of the fused adapter path (fixed-length batches): the largest power of two the LDS has room for
  bool neutral;        // strided batch whose pad bytes are 0xFF (QK_BATCH_NEUTRAL_PADS): the kernel variant without tail masks
};
constexpr unsigned kQueueRing = 8;       // queue sets that rotate (launches of one accumulator run in order)
constexpr unsigned kQueueTiles = 1u << 17;  // counters per launch: reads of up to 64 Mbases (512-position tiles)

// Launch geometry.  One position tile whenever the LDS histogram of the whole
// read fits (<= 576 positions, 448 with the adapter filter resident): tiles of
// one read live in different workgroups and share cache lines at the seams
// (measured on 300 bp: two 152-wide tiles 1.78 ms, one 304-wide tile 1.10 ms).
// Longer reads get tiles of <= 512 positions.  The grid is sized to residency:
// workgroups per CU = what the LDS admits (at most 2 of 1024 threads).
uint32_t single_tile_cap(const qk_accum *a, bool ragged, bool w16) {
  uint32_t cap = 576u;
  while (cap > 64u && qk::hist_lds_bytes(cap / 8, qk::hist_replicas(cap / 8), a->adapters, 0, ragged, qk::kStageReads, w16) > 160 * 1024)
    cap -= 32u;
  return cap;
}

// Grouped rows (qk::HistParams::group): how many consecutive reads of a fixed-length batch — `stride` bytes apart, a
// multiple of 4 — the fused adapter kernel should take as ONE row of 16-position lanes.  A row of G reads needs
// ceil(((G - 1) * stride + read_len) / 16) lanes; the G with the fewest lanes per base whose row still fits one tile
// wins (150 bp at stride 152: 10 lanes for one read, 19 for two; 100 bp: 7 / 19 for three; 36 bp: 3 / 9 for four), 1
// when nothing is gained.  Only under the planner's own geometry, with the first hits in the LDS ring.
uint32_t choose_group(const qk_accum *a, uint64_t n_reads, uint32_t read_len, uint32_t stride, bool base_aligned4, uint32_t row_cap = 0) {
  if (!a->adapters || getenv("QUACK_HIP_UNFUSED_ADAPTERS") || getenv("QUACK_HIP_NO_GROUP") || getenv("QUACK_HIP_NO_W16") ||
      getenv("QUACK_HIP_NO_ALIGN4") || tune_on("separate_count") || tune_on("adapt_pd") || tune_on("adapt_u"))
    return 1;
  if (a->unroll || a->pipe || a->threads != 1024 || a->tile > 0 || a->wgs_per_cu > 0) return 1;
  if ((stride & 3u) || stride < 16u || read_len < 11u || !base_aligned4) return 1;
  const uint32_t cap = single_tile_cap(a, false, true);   // positions of the widest one-tile row
  const int forced = tune_int("group", 0);       // (tests: a given group size, where it fits)
  uint32_t best = 1;
  double best_cost = 0;
  for (uint32_t g = 1; g <= 8u && g <= n_reads; ++g) {
    const uint32_t row = (g - 1u) * stride + read_len;
    if (g > 1u && ((uint32_t)round_up(row, 16) > cap || (row_cap && row > row_cap))) break;
    if (g > 1u && (1008u / (uint32_t)(round_up(row, 16) / 16)) * g > qk::kFhRing / 2u) continue;   // reads per step must fit half the ring
    const double cost = (double)round_up(row, 16) / ((double)g * read_len);
    if (forced > 0 ? g == (uint32_t)forced : (g == 1u || cost < best_cost * 0.985)) {
      best = g;
      best_cost = cost;
    }
  }
  return best;
}

int make_plan(const qk_accum *a, uint64_t n_reads, uint32_t max_len, bool ragged, bool gapped, bool aligned, Plan *pl,
              bool base_aligned4 = true, bool strided = false, uint32_t addr_stride = 0, bool neutral_req = false, bool sv_w16 = false) {
  const uint32_t T = (uint32_t)a->threads;
  // fixed-length reads: the distance between two reads — the read length, or (padded batches, round 4) the multiple of 4
  // above it: what decides whether every chunk starts on a dword is the stride, not the length
  const uint32_t fstride = addr_stride ? addr_stride : max_len;
  pl->fused_adapters = a->adapters && !getenv("QUACK_HIP_UNFUSED_ADAPTERS");
  // reads per lane and step / software pipeline depth.  Measured (10M x 150,
  // 5M x 300 + adapters, 1-20 kb ragged; kbench): fixed-length batches like
  // the next step's loads in flight while one is consumed — with one read per
  // step when there is no adapter scan (70 VGPRs, 0.546 -> 0.527 ms), two
  // with it (round 2, 10M x 300 spliced: (4,2) 1.542, (2,2) 1.495, (1,2) 1.495,
  // (4,1) 1.508, (1,1) 1.771 ms); the ragged path (LDS-staged descriptors) is
  // best unpipelined.
  const bool want_pipe = a->pipe > 0 ? a->pipe > 1 : (a->unroll <= 0 && !ragged && T == 1024);
  pl->pipe = a->pipe > 0 ? a->pipe : (want_pipe ? 2 : 1);
  pl->unroll = a->unroll > 0 ? a->unroll : ((want_pipe && !ragged) ? (pl->fused_adapters ? 2 : 1) : 4);
  // 16 positions per lane (W16): every batch form whose chunks are dword aligned — reads on cache lines in several
  // tiles, fixed-length reads of a multiple of 4 bases — under the planner's own geometry
  const bool tuned = a->unroll || a->pipe || T != 1024;
  // (measured, round 3, same box: 10M x 300 + adapters 1.385 -> 1.27-1.29 ms, 1-20 kb reads on cache lines 0.575 -> 0.56;
  // fixed-length reads WITHOUT the adapter scan are memory-bound with 8 positions per lane at 70 VGPRs and lose 2-9 %
  // with 16 at 107 — 100 bp 0.341 -> 0.354 ms, 36 bp 0.140 -> 0.154 —, so they keep one chunk per lane)
  // (strided batches — trimmed reads — take it too when their pads are neutral and the adapter scan is fused in: round 5)
  const bool neutral_ok = neutral_req && strided && !tuned && T == 1024 && !tune_on("length_kernel") && !getenv("QUACK_HIP_NO_NEUTRAL");
  bool w16 = !tuned && (!strided || (neutral_ok && sv_w16)) && !getenv("QUACK_HIP_NO_W16") &&
             (ragged ? aligned : (pl->fused_adapters && (fstride & 3u) == 0 && base_aligned4 && !getenv("QUACK_HIP_NO_ALIGN4") &&
                                  // short reads whose last pair would be half empty lose more lanes than the pairs save
                                  // (36 bp: 48 columns for 36 positions, 0.236 -> 0.251 ms; 76 bp 0.417 -> 0.383, 100 bp 0.521 -> 0.497)
                                  (max_len >= 64 || round_up(max_len, 16) == round_up(max_len, 8))));
  if (tune_on("w16_always") && !tuned && !strided && !ragged && (fstride & 3u) == 0 && base_aligned4) w16 = true;   // (tests: the plain fixed-length variant)
  // widest tile whose LDS image (histogram + adapter tables + staged read list) fits
  const uint32_t single_cap = single_tile_cap(a, ragged, w16);
  uint32_t cap = a->tile > 0 ? std::max<uint32_t>(8, (uint32_t)a->tile / 8 * 8) : single_cap;
  cap = std::min(cap, single_cap);
  uint32_t n_tiles = std::max<uint32_t>(1, (max_len + cap - 1) / cap);
  if (n_tiles > 1 && cap > 512) {
    cap = 512;
    n_tiles = (max_len + cap - 1) / cap;
  }
  const uint32_t lanes8 = pl->fused_adapters ? T / 64 * 62 : T;   // two feeder lanes per wave when fused
  if (cap / 8 + 2 > lanes8) {
    cap = (lanes8 - 2) * 8;
    n_tiles = (max_len + cap - 1) / cap;
  }
  uint32_t tile_pos = (uint32_t)round_up((max_len + n_tiles - 1) / n_tiles, 8);
  if (tile_pos == 0) tile_pos = 8;
  // reads that start on cache lines: tiles of whole cache lines (every line is
  // then fetched by exactly one workgroup); only worth it with several tiles
#ifdef QK_ABLATION
  pl->aligned = aligned && ragged && n_tiles > 1 && cap >= 128 && T == 1024;
#else
  pl->aligned = aligned && ragged && n_tiles > 1 && cap >= 128 && T == 1024 && !a->unroll && !a->pipe;
#endif
  if (pl->aligned) {
    tile_pos = cap / 128 * 128;
    n_tiles = (max_len + tile_pos - 1) / tile_pos;
    // cache-line tiles of long reads like two reads per lane and step with the next step's loads in
    // flight (1-20 kb reads: (4,1) 0.657 ms, (2,2) 0.631, (4,2) 0.636, (1,2) 0.651, (2,1) 0.701)
    if (!a->unroll && !a->pipe) {
      pl->unroll = 2;
      pl->pipe = 2;
    }
  }
  if (ragged && !pl->aligned) w16 = false;   // (a packed batch, or one tile: 12-byte windows)
  // fixed-length reads of a multiple of 4 bases: every chunk is dword aligned
  // (the batch base is: hipMalloc / pinned slots; submit_device checks it)
  if (!ragged && (fstride & 3u) == 0 && base_aligned4 && T == 1024 && !a->unroll && !a->pipe &&
      !getenv("QUACK_HIP_NO_ALIGN4"))
    pl->aligned = true;
  if (pl->aligned && !ragged && pl->fused_adapters) pl->pipe = tune_int("adapt_pd", pl->pipe), pl->unroll = tune_int("adapt_u", pl->unroll);
  if (!pl->aligned || (pl->fused_adapters && (tune_on("adapt_pd") || tune_on("adapt_u")))) w16 = false;
  if (w16) {
    // a lane owns two adjacent chunks: tiles of whole chunk pairs (fixed-length reads: the columns behind the
    // read hold the next read's bytes and are never flushed, as before), one read per lane and step with the
    // next step's loads in flight — the same bytes in flight as two reads of 8 positions
    tile_pos = (uint32_t)round_up(tile_pos, 16);
    pl->unroll = 1;
    pl->pipe = 2;
  }
  pl->w16 = w16;
  pl->n_tiles = n_tiles;
  pl->tile_pos = tile_pos;
  pl->ch = tile_pos / 8;
  // lanes covering the 16 positions before a tile: two chunks of 8 (W16: one lane)
  pl->halo = (pl->fused_adapters && n_tiles > 1) ? (w16 ? 1u : 2u) : 0u;
  // lanes per workgroup that own chunks (fused: two feeder lanes per wave, W16 one) / lanes per read row
  pl->rw = (pl->fused_adapters ? T / 64 * (w16 ? 63 : 62) : T) / (pl->ch / (w16 ? 2 : 1) + pl->halo);
  // replicas of the quality counters (bank balance, see qk::hist_replicas)
  pl->replicas = qk::hist_replicas(pl->ch);
  if (const int r = tune_int("replicas", 0)) pl->replicas = std::min<uint32_t>((uint32_t)r, pl->replicas);
  const uint64_t step = (uint64_t)pl->rw * (uint32_t)pl->unroll;
  // the exact table next to the histogram, if it fits (it never narrows a tile: without it
  // the queued candidates are checked against the global table)
  pl->bucket_log2 = pl->fused_adapters ? a->bucket_log2 : 0;
  // (the exact table matters more than the last replica)
  while (pl->bucket_log2 && pl->replicas > 1 && qk::hist_lds_bytes(pl->ch, pl->replicas, true, pl->bucket_log2, ragged, qk::kStageReads, w16) > 160 * 1024)
    --pl->replicas;
  if (pl->bucket_log2 && qk::hist_lds_bytes(pl->ch, pl->replicas, true, pl->bucket_log2, ragged, qk::kStageReads, w16) > 160 * 1024) pl->bucket_log2 = 0;
  // ragged: reads staged per pass.  A pass should hold many steps (36 bp reads: 816 per
  // step, so passes of 1024 staged every 1.25 steps: 1.10 ms per 40M reads, 0.5x the fixed
  // path), as far as the LDS next to the histogram allows
  pl->stage_reads = qk::kStageReads;
  if (ragged) {
    const uint64_t want = step * 12;
    while (pl->stage_reads < qk::kStageReadsMax && pl->stage_reads < want &&
           qk::hist_lds_bytes(pl->ch, pl->replicas, pl->fused_adapters, pl->bucket_log2, true, pl->stage_reads * 2, w16) <= 160 * 1024)
      pl->stage_reads *= 2;
  }
  // First-hit ring of the fused adapter path (fixed-length batches): the largest power of two the LDS has room for.  Every
  // fh_words / 2 reads the workgroup folds the ring — every wave checks its queue, however empty, and they meet at a barrier:
  // 10M x 150 + adapters, same box: 2048 words (a fold every 10 steps) 0.6474 ms, 8192 words 0.6268.  A ring of 8192 words is
  // worth more than counter replicas (which measure nothing with the adapter scan: 0.6479 against 0.6474 ms): they go first.
  pl->fh_words = qk::kFhRing;
  if (pl->fused_adapters && !ragged && !tune_on("small_ring")) {
    const uint32_t most = (uint32_t)std::max(2048, std::min((int)qk::kFhRingMax, tune_int("ring_words", (int)qk::kFhRingMax)));
    auto ring_for = [&](uint32_t replicas) {
      uint32_t wds = qk::kFhRing;
      while (wds * 2 <= most && qk::hist_lds_bytes(pl->ch, replicas, true, pl->bucket_log2, false, pl->stage_reads, w16, wds * 2) <= 160 * 1024)
        wds *= 2;
      return wds;
    };
    while (pl->replicas > 1 && ring_for(pl->replicas) < std::min<uint32_t>(most, 8192u) && !tune_on("replicas")) --pl->replicas;
    pl->fh_words = ring_for(pl->replicas);
  }
  size_t lds = qk::hist_lds_bytes(pl->ch, pl->replicas, pl->fused_adapters, pl->bucket_log2, ragged, pl->stage_reads, w16, pl->fh_words);
  if (lds > 160 * 1024) return fail(QK_EINVAL, "LDS tile too large (%zu bytes)", lds);
  // residency: the kernels need 89-104 VGPRs, i.e. 4 waves per SIMD = 1024
  // threads per CU, and the LDS image must fit as many times
  uint32_t wgs = a->wgs_per_cu > 0 ? (uint32_t)a->wgs_per_cu
                                   : std::max<uint32_t>(1, std::min<uint32_t>(1024 / T, (uint32_t)(160 * 1024 / lds)));
  // strided batches with neutral pads (0xFF behind every read): the variant without tail masks — one tile, lengths counted in
  // the step loop (which is where a position's `valid` count comes from), the planner's own geometry
  pl->neutral = neutral_ok && n_tiles == 1;
  // strided batches without the adapter scan: the kernel is built for 64 VGPRs and runs two
  // workgroups per CU when the LDS holds two histograms (reads of up to ~190 bases)
  if (strided && !pl->fused_adapters && a->wgs_per_cu <= 0 && T == 1024) {
    if (2 * lds > 160 * 1024 && !tune_on("sv_one_wg")) {   // fewer replicas, if that admits the second workgroup
      const size_t lds1 = qk::hist_lds_bytes(pl->ch, 1, false, 0, ragged, pl->stage_reads);
      if (2 * lds1 <= 160 * 1024) {
        pl->replicas = 1;
        lds = lds1;
      }
    }
    if (2 * lds <= 160 * 1024) wgs = 2;
  }
  // Work items = tiles x read slices.  Single tile: one item per resident
  // workgroup.  Several tiles (long reads): reads do not reach the far tiles
  // equally, so the slices are cut ~16x finer than the resident workgroups and
  // pulled from per-tile device counters (see hist_kernel: home tiles); a
  // workgroup flushes its LDS histogram only when its tile changes.
  const uint64_t resident = (uint64_t)a->n_cu * wgs;
  pl->dynamic = n_tiles > 1;
  uint64_t want_items = pl->dynamic ? resident * (uint64_t)std::max(1, tune_int("oversub", 16)) : resident;
  uint64_t n_slices = std::max<uint64_t>(1, want_items / n_tiles);
  uint64_t rps = (n_reads + n_slices - 1) / n_slices;
  rps = round_up(std::max<uint64_t>(rps, 1), step);
  // u16 LDS counters: a workgroup may see at most 65535 reads between flushes
  if (step * 2 > qk::kMaxReadsPerSlice) return fail(QK_EINVAL, "tile too wide for u16 counters");
  uint64_t rcap = (qk::kMaxReadsPerSlice - step) / step * step;
  // the kernel addresses a slice with 32-bit byte offsets
  // (a gapped batch is < 2 GiB as a whole, checked by the caller: any slice fits)
  if (!gapped) {
    const uint64_t by_bytes = (0x7FFFFFFFull / std::max<uint32_t>(ragged ? max_len : fstride, 1)) / step * step;
    if (by_bytes < step) return fail(QK_EINVAL, "reads of %u bytes are too long for one slice", max_len);
    rcap = std::min(rcap, by_bytes);
  }
  if (rps > rcap) {
    // more reads than one residency round may count in u16: whole rounds of
    // equal slices, so that the last round is not a nearly empty one
    const uint64_t per_round = (pl->dynamic ? 1 : resident) * rcap;
    const uint64_t rounds = (n_reads + per_round - 1) / per_round;
    n_slices = (pl->dynamic ? n_slices : resident) * rounds;
    rps = round_up((n_reads + n_slices - 1) / n_slices, step);
    if (rps > rcap) rps = rcap;
  }
  pl->reads_per_slice = rps;
  pl->n_slices = (n_reads + rps - 1) / rps;
  if (pl->n_slices > 0xFFFFFF00ull) return fail(QK_EINVAL, "too many read slices");
  pl->n_blocks = pl->dynamic ? resident : pl->n_slices;   // persistent workgroups when dynamic
  return QK_OK;
}

// Padded fixed-length batches (round 4).  Uniform reads whose length is not a multiple of 4 — 150, 250, 125, 50 bp —
// start on 2- or 1-byte phases when packed, so their chunks are fetched through 12-byte windows and re-aligned, and the
// 16-positions-per-lane kernel (one dwordx4 per lane and array) cannot take them.  Laid out at a stride rounded up to 4 the
// batch runs the dword-aligned kernels: with the adapter scan 150 bp went from 0.525 of the HBM peak to the 300-bp twin's
// range (profiles/r04_lengths.log) for 1.3 % more bytes.  Returns the stride a feed should use for `read_len`, 0: packed.
uint32_t padded_stride_for(const qk_accum *a, uint32_t read_len) {
  if ((read_len & 3u) == 0 || getenv("QUACK_HIP_NO_PAD")) return 0;
  if (a->unroll || a->pipe || a->threads != 1024 || getenv("QUACK_HIP_NO_ALIGN4")) return 0;   // (the aligned variants are the planner's own)
  const bool always = tune_on("pad_always");
  if (!a->adapters && !always) return 0;
  if (read_len < 16u && !always) return 0;
  return (read_len + 3u) & ~3u;
}


template <int T, int U, int PD>
int launch_hist_tu(const qk::HistParams &hp, bool fixed, int mode, bool adapt, bool aligned, bool strided, bool w16, dim3 grid,
                   size_t lds, hipStream_t st, bool neutral = false) {
  void (*k)(const qk::HistParams) = nullptr;
  if (w16 && strided) {
    // strided rows with neutral pads and the fused adapter scan (round 5): one tile, block b owns slice b
    if constexpr (T == 1024 && PD == 2 && U == 1)
      if (fixed && aligned && mode == 0 && adapt && neutral && hp.n_tiles == 1 && !hp.queue && !hp.static_split)
        k = qk::hist_kernel<T, U, true, 0, true, PD, true, true, true, true, true>;
    if (!k) return fail(QK_EINVAL, "the 16-positions-per-lane strided kernel is built for one-tile batches with neutral pads and adapters only");
  } else
  if (w16) {
    // 16 positions per lane: built for the step shapes the planner asks for
    if (aligned && mode == 0 && !strided) {
      if constexpr (T == 1024 && PD == 2 && U == 1) {
        const bool one = hp.n_tiles == 1 && !hp.queue && !hp.static_split;
        if (fixed && adapt) k = one ? qk::hist_kernel<T, U, true, 0, true, PD, true, false, true, false, true> : qk::hist_kernel<T, U, true, 0, true, PD, true, false, true>;
        else if (fixed) k = qk::hist_kernel<T, U, true, 0, false, PD, true, false, true>;
        else k = adapt ? qk::hist_kernel<T, U, false, 0, true, PD, true, false, true> : qk::hist_kernel<T, U, false, 0, false, PD, true, false, true>;
      }
    }
    // (three register sets — pipe 3 — were built and measured twice on 10M x 300 + adapters: with spills 1.370 -> 1.427 ms;
    // after the tail masks went (no spills) 1.31-1.36 against 1.32-1.37 ms, inside the box's own scatter)
    if (!k) return fail(QK_EINVAL, "the 16-positions-per-lane kernel is not built for unroll %d / pipe %d", U, PD);
  } else
  if (strided) {
    // fixed stride + per-read lengths: built for the planner's own choice only
    if (fixed && aligned && mode == 0) {
      if constexpr (T == 1024 && PD == 2 && U == 1) {
        if (!adapt) k = neutral ? qk::hist_kernel<T, U, true, 0, false, PD, true, true, false, true> : qk::hist_kernel<T, U, true, 0, false, PD, true, true>;
      }
      if constexpr (T == 1024 && PD == 2 && U == 2) {
        if (adapt) k = neutral ? qk::hist_kernel<T, U, true, 0, true, PD, true, true, false, true> : qk::hist_kernel<T, U, true, 0, true, PD, true, true>;
      }
    }
    if (!k) return fail(QK_EINVAL, "strided batches run with the planner's own launch geometry only (threads/unroll/pipe overrides are set)");
  } else
  if (aligned && !fixed && mode == 0) {
    // only built for the planner's own choice (make_plan sets `aligned` for nothing else)
    if constexpr (T == 1024 && U == 2 && PD == 2)
      k = adapt ? qk::hist_kernel<T, U, false, 0, true, PD, true> : qk::hist_kernel<T, U, false, 0, false, PD, true>;
#ifdef QK_ABLATION   /* kbench / experiments: other step shapes of the cache-line variant */
    if constexpr (T == 1024 && ((U == 4 && PD == 1) || (U == 4 && PD == 2) || (U == 1 && PD == 2) || (U == 2 && PD == 1)))
      if (!adapt) k = qk::hist_kernel<T, U, false, 0, false, PD, true>;
#endif
  }
  // The product library holds the variants the planner itself asks for (make_plan without overrides): fixed length (1,2) plain and
  // (2,2) with the adapter scan, packed ragged (4,1), reads on cache lines (2,2), 16 positions per lane (1,2), strided.  Every other
  // step shape belongs to the -DQK_EXPERIMENT build (QUACK_HIP_TUNE, qk_accum_configure).
#ifdef QK_EXPERIMENT
  constexpr bool kAll = true;
#else
  constexpr bool kAll = false;
#endif
  if (!w16 && !strided && aligned && fixed && mode == 0) {
    if constexpr (T == 1024 && PD == 2 && (U == 1 || U == 2)) {
      if constexpr (kAll || U == 2)
        if (adapt) k = qk::hist_kernel<T, U, true, 0, true, PD, true>;
      if constexpr (kAll || U == 1)
        if (!adapt) k = qk::hist_kernel<T, U, true, 0, false, PD, true>;
    }
    if constexpr (kAll && T == 1024 && PD > 2)
      if (adapt) k = qk::hist_kernel<T, U, true, 0, true, PD, true>;
  }
  if constexpr (PD > 2) {
    if (!k) return fail(QK_EINVAL, "pipeline depth %d is built for dword-aligned fixed-length batches with adapters only", PD);
  } else
  if (k) {
  } else if (adapt) {
    if (mode == 0) {
      if constexpr (kAll || (T == 1024 && U == 2 && PD == 2))
        if (fixed) k = qk::hist_kernel<T, U, true, 0, true, PD>;
      if constexpr (kAll || (T == 1024 && U == 4 && PD == 1))
        if (!fixed) k = qk::hist_kernel<T, U, false, 0, true, PD>;
    }
  } else if (fixed) {
    switch (mode) {
      case 0:
        if constexpr (kAll || (T == 1024 && U == 1 && PD == 2)) k = qk::hist_kernel<T, U, true, 0, false, PD>;
        break;
#ifdef QK_ABLATION
      case 1: k = qk::hist_kernel<T, U, true, 1, false, PD>; break;
      case 2: k = qk::hist_kernel<T, U, true, 2, false, PD>; break;
      case 3: k = qk::hist_kernel<T, U, true, 3, false, PD>; break;
#endif
    }
  } else {
    if constexpr (kAll || (T == 1024 && U == 4 && PD == 1))
      if (mode == 0) k = qk::hist_kernel<T, U, false, 0, false, PD>;
  }
  if (!k) return fail(QK_EINVAL, "kernel variant not built: threads %d, unroll %d, pipe %d%s%s (mode %d) — the product library holds the "
                      "planner's own step shapes, the experiment build (libquack_hip_exp.so, QUACK_HIP_TUNE) the others", T, U, PD,
                      fixed ? ", fixed length" : ", ragged", adapt ? ", adapters" : "", mode);
  {
    // the kernels use absolute LDS addresses (qk::qhist_add; the fused scan's filter at
    // LDS byte 0, qk::lds_abs_u8): the dynamic segment must start at byte 0
    static bool checked = false;   // per instantiation
    if (!checked) {
      hipFuncAttributes fa;
      QK_HIP(hipFuncGetAttributes(&fa, (const void *)k));
      if (fa.sharedSizeBytes != 0) return fail(QK_ESTATE, "histogram kernel has %zu bytes of static LDS", (size_t)fa.sharedSizeBytes);
      checked = true;
    }
  }
  // The dynamic-LDS ceiling is a property of the function, shared by every thread and
  // accumulator of the process: always the full 160 KiB, never "what this launch needs"
  // (the two mates of a pair are accumulated by two host threads, and reads of different
  // lengths would otherwise lower each other's ceiling between attribute and launch).
  // ... and once per function and device, under a lock: the runtime call rewrites function
  // metadata that a concurrent launch of the same function reads.
  {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    QK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (!done.count({(const void *)k, dev})) {
      QK_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done.insert({(const void *)k, dev});
    }
  }
  static bool first_launch = true;   // (QUACK_VERBOSE: the first launch of the process loads the code object)
  Lap lap;
  hipLaunchKernelGGL(k, grid, dim3(T), lds, st, hp);
  QK_HIP(hipGetLastError());
  if (first_launch && lap.on) {
    first_launch = false;
    lap.mark("enqueue");
    lap.print("first histogram launch");
  }
  return QK_OK;
}

int launch_hist(qk_accum *a, const qk::HistParams &hp, const Plan &pl, bool fixed, int mode, bool adapt,
                hipStream_t st, bool strided = false) {
  const bool neutral = strided && pl.neutral;
  const size_t lds = qk::hist_lds_bytes(hp.ch, pl.replicas, adapt, hp.bucket_log2, !fixed, pl.stage_reads, pl.w16, pl.fh_words);
  dim3 grid((unsigned)pl.n_blocks);
#define QK_TU(TT, UU, PP) \
  if (a->threads == TT && pl.unroll == UU && pl.pipe == PP) return launch_hist_tu<TT, UU, PP>(hp, fixed, mode, adapt, pl.aligned, strided, pl.w16, grid, lds, st, neutral);
  QK_TU(1024, 4, 1) QK_TU(1024, 2, 2) QK_TU(1024, 1, 2)
#ifdef QK_EXPERIMENT
  QK_TU(1024, 2, 1) QK_TU(1024, 1, 1) QK_TU(1024, 4, 2)
  QK_TU(1024, 2, 3) QK_TU(1024, 2, 4) QK_TU(1024, 1, 4)
  QK_TU(512, 4, 1) QK_TU(512, 2, 1) QK_TU(512, 1, 1)
  QK_TU(256, 4, 1) QK_TU(256, 2, 1)
#endif
#undef QK_TU
  return fail(QK_EINVAL, "unsupported threads/unroll/pipe %d/%d/%d (the product library holds the planner's own step shapes; "
              "others: the experiment build)", a->threads, pl.unroll, pl.pipe);
}

hipEvent_t get_event(qk_accum *a) {
  if (!a->event_pool.empty()) {
    hipEvent_t e = a->event_pool.back();
    a->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

int g_ablation_mode = 0;  // set through qk_debug_set_mode (kbench only)

// Launches of one accumulator run in submission order even when they come from different streams (they share the
// queue ring, the first-hit scratch and the table's flush targets).  The event that orders them is recorded only
// when the stream actually changes — on the accumulator's own streams, which outlive the question; a caller's stream
// gets it right behind the launch, while it is known to exist.  (Round 2 recorded it behind every batch: a marker
// packet between any two launches of a device-resident loop, 3-5 us of the ~14 us between two kernels.)
int order_after_previous(qk_accum *a, hipStream_t st) {
  if (a->order_valid && a->order_stream != st) {
    if (!a->order_recorded) QK_HIP(hipEventRecord(a->order_ev, a->order_stream));
    a->order_recorded = true;
    QK_HIP(hipStreamWaitEvent(st, a->order_ev, 0));
  }
  return QK_OK;
}

// Enqueue the kernels of one device-resident batch on `st`.
int enqueue_batch(qk_accum *a, const uint8_t *d_seq, const uint8_t *d_qual,
                  const uint64_t *d_off, uint32_t *d_hit, uint64_t n_reads,
                  uint64_t total_bytes, uint32_t max_len, hipStream_t st,
                  const uint32_t *d_len = nullptr, uint32_t flags = 0, uint32_t stride = 0, bool no_group = false) {
  if (n_reads == 0) return QK_OK;
  if (n_reads > 0xFFFFFFF0ull) return fail(QK_EINVAL, "batch too large");
  int rc = grow_table(a, std::max<uint64_t>(max_len, 11));
  if (rc) return rc;
  if (max_len == 0) {  // only empty reads: they count as sequences, nothing else
    a->n_reads += n_reads;
    return QK_OK;
  }
  Plan pl;
  // batch forms: packed (d_off, no d_len), gapped (d_off = starts, d_len), strided (no d_off, d_len, stride), fixed length
  // (none of them; stride == 0 or max_len: read r at r * max_len) and PADDED fixed length (stride > max_len, a multiple of 4)
  const bool strided = d_off == nullptr && d_len != nullptr;
  const bool padded = d_off == nullptr && d_len == nullptr && stride > max_len;
  if (padded && (stride & 3u)) return fail(QK_EINVAL, "a padded stride must be a multiple of 4");
  if (d_off && d_len && total_bytes > 0x7FFFFFF0ull) return fail(QK_EINVAL, "a gapped batch must stay below 2 GiB");
  const bool base4 = (((uintptr_t)d_seq | (uintptr_t)d_qual) & 3u) == 0;
  // fixed-length reads with the fused adapter scan: rows of several reads where that fills the 16-position lanes better
  // (qk::HistParams::group); the reads that do not make a whole row run as a small batch of their own
  uint32_t group = 1;
  const uint32_t rstride = (strided || padded) ? stride : max_len;   // bytes between two reads of a fixed-stride batch
  // (round 5) strided batches whose pads are neutral take the same kernel when the adapter scan is fused in — trimmed reads run
  // with -a as a rule (/root/reference/images/makefile:8,14): a row is then whole strides, pads included
  const bool sv16 = strided && (flags & QK_BATCH_NEUTRAL_PADS) && a->adapters && (stride & 3u) == 0 && base4 && stride >= 16u &&
                    stride <= single_tile_cap(a, false, true);
  const uint32_t glen = sv16 ? stride : max_len;   // positions of a read's slot that the kernel counts
  if (!d_off && (!d_len || sv16) && !no_group && n_reads >= 64) {
    uint32_t row_cap = 0;
    for (;;) {
      group = choose_group(a, n_reads, glen, rstride, base4, row_cap);
      if (group <= 1) break;
      const uint32_t row = (group - 1u) * rstride + glen;
      rc = make_plan(a, n_reads / group, row, false, false, false, &pl, base4, sv16, group * rstride, sv16, sv16);
      if (rc == QK_OK && pl.w16 && pl.n_tiles == 1 && pl.fused_adapters && (uint64_t)pl.rw * (uint32_t)pl.unroll * group <= qk::kFhRing / 2u &&
          (pl.bucket_log2 || !a->bucket_log2) && (!sv16 || pl.neutral))
        break;
      row_cap = row - 1u;   // (the row does not fit beside the adapter tables after all: a shorter one)
    }
  }
  uint64_t rem = 0, rem_head = 0;   // the reads that do not fill a last row: a small launch of their own, inside this batch's timing events (below)
  if (group > 1) {
    rem = n_reads % group;
    if (rem) {
      rem_head = n_reads - rem;
      n_reads = rem_head;
      total_bytes = rem_head * (uint64_t)rstride;
    }
  } else {
    rc = make_plan(a, n_reads, strided ? stride : max_len, d_off != nullptr, d_off && d_len,
                   d_off && d_len && (flags & QK_BATCH_ALIGNED128), &pl, base4, strided, padded ? stride : 0u,
                   strided && (flags & QK_BATCH_NEUTRAL_PADS), sv16);
    if (rc) return rc;
    if (sv16 && pl.w16 && (pl.n_tiles != 1 || !pl.neutral || (uint64_t)pl.rw * (uint32_t)pl.unroll > qk::kFhRing / 2u)) {
      // (not the shape the 16-position strided kernel is built for after all)
      rc = make_plan(a, n_reads, stride, false, false, false, &pl, base4, true, 0u, true, false);
      if (rc) return rc;
    }
  }
  // QK_BATCH_NEUTRAL_PADS on a device-resident batch: the promise is verified for the first batches of every accumulator (a producer
  // that writes its pads wrongly does so from the start) and for every batch on request (QUACK_HIP_CHECK_PADS=1: tests,
  // debugging a producer) — whichever kernel variant ends up running the batch: the promise is about the data
  if (strided && (flags & QK_BATCH_NEUTRAL_PADS) && !no_group && (a->pads_checked < 2 || getenv("QUACK_HIP_CHECK_PADS"))) {
    a->pads_checked++;
    if ((rc = order_after_previous(a, st))) return rc;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_reads + rem + 255) / 256, 4096);
    hipLaunchKernelGGL(qk::pads_check_kernel, dim3(blocks), dim3(256), 0, st, d_seq, d_qual, d_len, n_reads + rem, stride, a->d_status);
    QK_HIP(hipGetLastError());
    a->status_armed = true;
  }
  if (strided) {
    // The strided kernel variant exists for the planner's own geometry only.  Under a tuning override
    // (QUACK_HIP_THREADS / _UNROLL / _PIPE / _NO_ALIGN4 / _ADAPT_PD / _ADAPT_U, qk_accum_configure) the same
    // reads run as gapped batches — starts[i] = i * stride written by a small kernel, lengths[] as they are —
    // in chunks below the 2 GiB a gapped batch may span.  Same counters, the ragged kernels' speed.
    const bool native = pl.aligned && a->threads == 1024 && pl.pipe == 2 && pl.unroll == ((pl.fused_adapters && !pl.w16) ? 2 : 1);
    if (!native) pl.neutral = false;
    if (!native) {
      if (a->starts_scratch_reads < n_reads) {
        QK_HIP(hipDeviceSynchronize());
        if (a->d_starts_scratch) QK_HIP(hipFree(a->d_starts_scratch));
        a->d_starts_scratch = nullptr;
        a->starts_scratch_reads = 0;
        QK_HIP(hipMalloc((void **)&a->d_starts_scratch, n_reads * sizeof(unsigned long long)));
        a->starts_scratch_reads = n_reads;
      }
      if ((rc = order_after_previous(a, st))) return rc;   // the scratch is shared by the accumulator's launches
      uint64_t per_chunk = std::max<uint64_t>(1, 0x7FFFFF00ull / stride);
      if (const int t = tune_int("strided_chunk_reads", 0)) per_chunk = std::min<uint64_t>(per_chunk, (uint64_t)t);   // (tests: several chunks without 2 GiB)
      const unsigned blocks = (unsigned)std::min<uint64_t>((n_reads + 255) / 256, 4096);
      hipLaunchKernelGGL(qk::strided_starts_kernel, dim3(blocks), dim3(256), 0, st, a->d_starts_scratch, n_reads, per_chunk, stride,
                         d_len, max_len, a->d_status);
      QK_HIP(hipGetLastError());
      a->status_armed = true;
      for (uint64_t lo = 0; lo < n_reads; lo += per_chunk) {
        const uint64_t cnt = std::min<uint64_t>(per_chunk, n_reads - lo);
        rc = enqueue_batch(a, d_seq + lo * stride, d_qual + lo * stride, (const uint64_t *)a->d_starts_scratch + lo,
                           d_hit ? d_hit + lo : nullptr, cnt, cnt * (uint64_t)stride, max_len, st, d_len + lo, 0, 0);
        if (rc) return rc;
      }
      return QK_OK;
    }
  }
  if ((rc = order_after_previous(a, st))) return rc;
  TimedLaunch tl{};
  // (events around a launch cost ~10 us of stream time: a caller that also measures its own wall
  // clock asks for every Nth batch only)
  // (the few reads that did not make a whole row of a grouped batch are not a launch of their own to the timing hooks)
  const bool timed = a->timing > 0 && !no_group && (a->timing_seq++ % (uint64_t)a->timing) == 0;
  if (timed) {
    tl.t0 = get_event(a);
    tl.t1 = get_event(a);
    if (!tl.t0 || !tl.t1) return fail(QK_EHIP, "hipEventCreate failed");
    tl.b0 = tl.t0;
    tl.b1 = tl.t1;
    // batches with a pre-pass (queue reset, reach sort, length kernel) or an adapter
    // count kernel get their own pair of events around the whole batch
    // (not a fused one-tile adapter batch: the histogram kernel is all of it, and two more events would add their own
    // ~10 us of stream time to what they measure)
    const bool one_kernel = pl.fused_adapters && pl.n_tiles == 1 && !tune_on("separate_count");
    if (pl.n_tiles > 1 || (a->adapters && !one_kernel) || (strided && tune_on("length_kernel"))) {
      tl.b0 = get_event(a);
      // (nothing runs behind the histogram kernel unless the adapter hits are counted by kernels of their own: the batch
      // then ends where the kernel does, one event for both)
      if (a->adapters && !one_kernel) tl.b1 = get_event(a);
      if (!tl.b0 || !tl.b1) return fail(QK_EHIP, "hipEventCreate failed");
      QK_HIP(hipEventRecord(tl.b0, st));
    }
  }
  qk::HistParams hp{};
  hp.seq = d_seq;
  hp.qual = d_qual;
  hp.offsets = d_off;
  hp.lengths = d_len;
  hp.stage_reads = pl.stage_reads;
  hp.status = a->d_status;
  hp.check_aligned = (d_off && d_len && (flags & QK_BATCH_ALIGNED128)) ? 1u : 0u;
  hp.len_limit = strided ? max_len : 0u;
  if (hp.check_aligned || strided) a->status_armed = true;   // (strided: the length kernel vets lengths[])
  hp.table = a->d_table;
  hp.table32 = getenv("QUACK_HIP_NO_TABLE32") ? nullptr : a->d_table32;
  if (hp.table32) {
    // a 32-bit counter takes at most one count per read and position: fold before 2^32 reads could have gone in
    if (a->reads32 + n_reads > 0xFFFFFFFFull && (rc = fold_table32(a, st))) return rc;
    a->reads32 += n_reads;
  }
  hp.no_adapters = a->adapters ? 0 : 1;
  hp.first_hit = d_hit;
  hp.kmer_bits = a->d_kmer_bits;
  hp.kmer_filter = a->d_kmer_filter;
  hp.kmer_buckets = a->d_kmer_buckets;
  hp.bucket_log2 = pl.bucket_log2;
  hp.bucket_mul = a->bucket_mul;
  hp.n_reads = n_reads / group;   // (rows)
  hp.fh_words = pl.fh_words;
  hp.group = group;
  hp.gstride = rstride;
  hp.total_bytes = total_bytes;
  hp.reads_per_slice = pl.reads_per_slice;
  hp.read_len = d_off ? 0 : (strided ? stride : max_len);
  hp.stride = d_off ? 0 : rstride * group;   // (between rows)
  hp.table_len = (uint32_t)a->table_len;
  hp.n_tiles = pl.n_tiles;
  hp.tile_pos = pl.tile_pos;
  hp.ch = pl.ch;
  hp.reads_per_iter = pl.rw;
  hp.n_slices = (uint32_t)pl.n_slices;
  hp.queue = nullptr;
  // (decided here, used below: reads sorted by reach; up to 64 tiles the read-tiles are split statically)
  pl.sorted = pl.dynamic && d_off != nullptr && total_bytes < 0xFFFFFF00ull && pl.n_tiles >= 4 &&
              pl.n_tiles <= qk::kReachMaxTiles && !getenv("QUACK_HIP_NO_SORT");
  const bool static_split = pl.dynamic && pl.sorted && pl.n_tiles <= 64 && !getenv("QUACK_HIP_NO_STATIC");
  if (pl.dynamic && !static_split) {
    if (pl.n_tiles > kQueueTiles) return fail(QK_EINVAL, "reads of %u bytes need more position tiles than supported", max_len);
    hp.queue = a->d_queues + (size_t)(a->queue_seq++ % kQueueRing) * kQueueTiles;
    QK_HIP(hipMemsetAsync(hp.queue, 0, pl.n_tiles * sizeof(uint32_t), st));
  }
  // several tiles, ragged: order the reads by the tiles they reach first, so that a
  // far tile only looks at the reads that get there (config 5: most staging passes
  // of the far tiles found next to nothing)
  hp.order = nullptr;
  hp.reach = nullptr;
  hp.lengths_done = 0;
  // (with two or three tiles nearly every read reaches every tile: nothing to gain from sorting)
  // kernels that only read the batch's lengths go to the side stream when the batch came in on the accumulator's own
  // stream (see qk_accum::side)
  const bool aside = st == a->stream && a->side != nullptr && !getenv("QUACK_HIP_NO_SIDE");
  hipStream_t pre = aside ? a->side : st;
  int set = -1;
  // (one tile, the strided kernel variant itself: the lane that owns a read's first chunk counts its length inside the step
  // loop — the separate pass over lengths[] was 24 us per 10M reads, 4.5 % of the batch)
  const bool count_in_loop = strided && pl.n_tiles == 1 && !tune_on("length_kernel");
  if (strided && !count_in_loop) {
    // strided batches have no staging pass that could count the lengths on the way
    // (ragged batches of several tiles: hist_kernel counts a read's length in the tile it ends in).
    // (on the side stream it would run BESIDE this batch's histogram kernel — nothing orders the two — and was measured:
    // the step 1 % shorter, the histogram kernel's own time 3 % longer, 0.540 -> 0.555-0.56 ms; it stays in line)
    const unsigned lb = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_reads / 8192, (uint64_t)a->n_cu));
    if (strided && stride <= qk::kShortLen)
      hipLaunchKernelGGL(qk::short_length_kernel, dim3(lb), dim3(qk::kLenThreads), 0, st, hp);
    else
      hipLaunchKernelGGL(qk::ragged_length_kernel, dim3(lb), dim3(qk::kLenThreads), 0, st, hp);
    QK_HIP(hipGetLastError());
    hp.lengths_done = 1;
  }
  if (pl.sorted) {
    set = (int)(a->set_turn++ % (unsigned)kSets);
    if (a->order_cap[set] < n_reads) {
      QK_HIP(hipDeviceSynchronize());
      if (a->d_order[set]) QK_HIP(hipFree(a->d_order[set]));
      a->d_order[set] = nullptr;
      a->order_cap[set] = 0;
      QK_HIP(hipMalloc((void **)&a->d_order[set], n_reads * sizeof(uint32_t)));
      a->order_cap[set] = n_reads;
    }
    // reach | counts | cursor | done | prefix (u64, 8-byte aligned: the word count before it is even)
    const size_t reach_words = 3 * (size_t)qk::kReachMaxTiles + 4 + 2 * ((size_t)qk::kReachMaxTiles + 1);
    if (!a->d_reach[set]) {
      QK_HIP(hipMalloc((void **)&a->d_reach[set], reach_words * sizeof(uint32_t)));
      QK_HIP(hipMemset(a->d_reach[set], 0, reach_words * sizeof(uint32_t)));   // counts and `done` stay zero between launches
    }
    uint32_t *reach = a->d_reach[set], *counts = reach + qk::kReachMaxTiles, *cursor = counts + qk::kReachMaxTiles + 1;
    uint32_t *done = cursor + qk::kReachMaxTiles + 1;
    unsigned long long *prefix = reinterpret_cast<unsigned long long *>(reach + 3 * (size_t)qk::kReachMaxTiles + 4);
    // the set's previous user — the histogram kernel of two batches ago — must have finished with it
    if (a->set_busy[set]) QK_HIP(hipStreamWaitEvent(pre, a->set_free[set], 0));
    const size_t lds = (pl.n_tiles + 2) * sizeof(uint32_t);
    // few blocks: every block costs one same-address atomic per bucket (~15 ns each, serialised)
    const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_reads / 1024, (uint64_t)a->n_cu * 4));
    hipLaunchKernelGGL(qk::reach_count_kernel, dim3(blocks), dim3(qk::kReachThreads), lds, pre, hp, counts, done, reach, cursor, prefix);
    hipLaunchKernelGGL(qk::reach_scatter_kernel, dim3(blocks), dim3(qk::kReachThreads), lds, pre, hp, cursor, a->d_order[set]);
    QK_HIP(hipGetLastError());
    if (aside) {
      QK_HIP(hipEventRecord(a->side_done, a->side));
      QK_HIP(hipStreamWaitEvent(st, a->side_done, 0));
    }
    hp.order = a->d_order[set];
    hp.reach = reach;
    hp.tile_prefix = prefix;
  }
  // several tiles and every tile's work known (reads sorted by reach): equal static shares instead of
  // the slice queues; an item is then as large as the u16 counters allow.  (Fixed-length long reads know
  // their work too, but measured SLOWER this way, 0.60 -> 0.66 ms per 143k x 10.5 kb: their workgroups
  // then march through equally strided addresses in lockstep; the queues desynchronise them.)
  // (And beyond ~32 kb — more than 64 tiles — the queues win again: 30k x 10-50 kb 0.459 static vs 0.435,
  // 3000 x 100-500 kb 1.18 vs 0.95; up to there static wins: 300k x 0.1-5 kb 0.355 vs 0.410, config 5.)
  if (static_split) {
    hp.static_split = 1;
    // what crossing into another tile costs a workgroup (clearing and flushing a histogram), in read-tiles: measured on
    // 1-20 kb reads inside one process (tools/ab_inproc.py): 0 -> 0.5822 ms, 150 0.5812, 300 0.5738, 450 0.5746,
    // 600 0.5737, 900 0.5755, 1200 0.5766, 2400 0.5815, 4800 0.5835
    hp.tile_overhead = (uint32_t)std::max(0, tune_int("tile_overhead", 400));
    const uint64_t step = (uint64_t)pl.rw * (uint32_t)pl.unroll;
    hp.reads_per_slice = (qk::kMaxReadsPerSlice - step) / step * step;
  }
  hp.row_dwords = qk::hist_row_dwords(pl.ch, pl.replicas);
  hp.replicas = pl.replicas;
  hp.halo = pl.halo;

  // one tile: the histogram kernel resets first_hit[] and takes the kmer_count of its own reads
  hp.count_in_kernel = pl.fused_adapters && pl.n_tiles == 1 && !tune_on("separate_count");
  // (Round 3 had an opt-in that handed the two events to the dispatch packet itself — hipExtLaunchKernelGGL, 0.5 % of a step —
  // and one of three full test runs with it stopped making progress; never root-caused, so round 4 removed it.)
  if (timed) QK_HIP(hipEventRecord(tl.t0, st));
  if (rem) {   // (this launch used to sit outside the events while its reads were in the counters)
    rc = enqueue_batch(a, d_seq + rem_head * rstride, d_qual + rem_head * rstride, nullptr, d_hit ? d_hit + rem_head : nullptr, rem, rem * (uint64_t)rstride,
                       max_len, st, d_len ? d_len + rem_head : nullptr, d_len ? flags : 0u, stride, /*no_group=*/true);
    if (rc) return rc;
  }
  if (pl.fused_adapters && !hp.count_in_kernel) QK_HIP(hipMemsetAsync(d_hit, 0xFF, n_reads * sizeof(uint32_t), st));
  rc = launch_hist(a, hp, pl, d_off == nullptr, g_ablation_mode, pl.fused_adapters, st, strided);
  if (rc) return rc;
  if (timed) QK_HIP(hipEventRecord(tl.t1, st));
  if (a->adapters && !hp.count_in_kernel) {
    // fused: the histogram pass already left first_hit[]; otherwise scan now
    rc = pl.fused_adapters ? qk::launch_adapter_count(hp, a->n_cu, st) : qk::launch_adapter_scan(hp, a->n_cu, st);
    if (rc) return fail(QK_EHIP, "adapter kernels failed: %s", hipGetErrorString((hipError_t)rc));
  }
  if (timed) {
    if (tl.b1 != tl.t1) QK_HIP(hipEventRecord(tl.b1, st));
    a->timed.push_back(tl);
  }
  if (set >= 0) {   // the next user of this set of order / reach buffers waits for the kernels above
    QK_HIP(hipEventRecord(a->set_free[set], st));
    a->set_busy[set] = true;
  }
  {
    bool own = st == a->stream;
    for (int i = 0; i < 2; ++i) own = own || (a->slot[i].stream && st == a->slot[i].stream);
    a->order_recorded = false;
    if (!own) {   // a caller's stream: it may be gone by the time another stream asks
      QK_HIP(hipEventRecord(a->order_ev, st));
      a->order_recorded = true;
    }
  }
  a->order_stream = st;
  a->order_valid = true;
  a->n_reads += n_reads;
  a->max_len = std::max<uint64_t>(a->max_len, max_len);
  return QK_OK;
}

int drain_timing(qk_accum *a) {
  for (auto &tl : a->timed) {
    float ms = 0, bms = 0;
    QK_HIP(hipEventSynchronize(tl.b1));
    QK_HIP(hipEventElapsedTime(&ms, tl.t0, tl.t1));
    bms = ms;
    if (tl.b0 != tl.t0) QK_HIP(hipEventElapsedTime(&bms, tl.b0, tl.b1));
    a->timing_ms += ms;
    a->timing_batch_ms += bms;
    if (a->timing_launches == 0 || ms < a->timing_min_ms) a->timing_min_ms = ms;
    if (a->timing_launches == 0 || ms > a->timing_max_ms) a->timing_max_ms = ms;
    a->timing_launches += 1;
    a->event_pool.push_back(tl.t0);
    a->event_pool.push_back(tl.t1);
    if (tl.b0 != tl.t0) a->event_pool.push_back(tl.b0);
    if (tl.b1 != tl.t1) a->event_pool.push_back(tl.b1);
  }
  a->timed.clear();
  return QK_OK;
}

int set_device(const qk_accum *a) {
  QK_HIP(hipSetDevice(a->device));
  return QK_OK;
}

}  // namespace

// ---- RCCL (loaded on first use; libquack_hip.so itself does not link it) ----
// The two mates of a pair are accumulated by two host threads (cli.c), and both
// end in qk_accum_allreduce over the same devices.  Everything RCCL — loading the
// library, creating communicators, the collective and the wait for it — therefore
// runs under ONE process-wide lock: two ncclCommInitAll over the same devices at
// the same time, or two collectives on different communicators enqueued in a
// different order on different devices, are the classic multi-communicator hangs.
// Communicators are created once per device set and kept for the life of the
// process (a file pair would otherwise pay the ~100 ms of ncclCommInitAll twice).
// <rccl/rccl.h> is included for its DECLARATIONS only — librccl.so is still loaded with dlopen on first use,
// so a single-GPU run never maps it — and the values and signatures the dlsym'd pointers below are called
// with are checked against it at compile time.
#include <rccl/rccl.h>
#include <type_traits>
namespace {
struct RcclApi {
  int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)(void) = nullptr;
  int (*GroupEnd)(void) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int *) = nullptr;
  char why[256] = "";
  bool ok = false;
};
constexpr int kNcclUint64 = 5, kNcclSum = 0;   // rccl.h: ncclUint64 = 5 (ncclDataType_t), ncclSum = 0 (ncclRedOp_t)
static_assert((int)ncclUint64 == kNcclUint64 && (int)ncclSum == kNcclSum && (int)ncclSuccess == 0,
              "RCCL enum values changed: the all-reduce of the counter tables would run with the wrong type or operator");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int),
              "RCCL enums are no longer int-sized: the dlsym'd signatures below pass them as int");
static_assert(std::is_same<decltype(&ncclAllReduce), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>::value &&
              std::is_same<decltype(&ncclCommInitAll), ncclResult_t (*)(ncclComm_t *, int, const int *)>::value &&
              std::is_same<decltype(&ncclGetErrorString), const char *(*)(ncclResult_t)>::value &&
              std::is_same<decltype(&ncclGetVersion), ncclResult_t (*)(int *)>::value,
              "RCCL prototypes differ from the ones rccl_api() casts its dlsym results to");

RcclApi g_rccl;
std::once_flag g_rccl_once;
std::mutex g_rccl_mu;                                      // serialises every RCCL section of the process
std::map<std::vector<int>, std::vector<ncclComm_t>> g_rccl_comms;   // device set -> communicators (guarded by g_rccl_mu)

const RcclApi &rccl_api() {
  std::call_once(g_rccl_once, [] {
    void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
      const char *e = dlerror();
      snprintf(g_rccl.why, sizeof g_rccl.why, "cannot load librccl.so: %s", e ? e : "?");
      return;
    }
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(lib, "ncclCommInitAll");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(lib, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(lib, "ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    g_rccl.GetVersion = (decltype(g_rccl.GetVersion))dlsym(lib, "ncclGetVersion");   // (optional: diagnostics only)
    if (!g_rccl.CommInitAll || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.GetErrorString) {
      snprintf(g_rccl.why, sizeof g_rccl.why, "librccl.so lacks a required symbol");
      return;
    }
    g_rccl.ok = true;
  });
  return g_rccl;
}

// RCCL writes its version banner and NCCL_DEBUG output to STDOUT — which for
// quack is the SVG document (measured: "RCCL version ..." in front of <svg> with
// QUACK_DEVICES=0,1).  While the RCCL section runs (under g_rccl_mu), file
// descriptor 1 points at stderr; stdio buffers are flushed on both sides of
// the switch so that nothing crosses over.
struct StdoutToStderr {
  int saved = -1;
  StdoutToStderr() {
    fflush(stdout);
    saved = dup(1);
    if (saved >= 0 && dup2(2, 1) < 0) {
      close(saved);
      saved = -1;
    }
  }
  ~StdoutToStderr() {
    if (saved < 0) return;
    fflush(stdout);
    (void)dup2(saved, 1);
    close(saved);
  }
};

// One all-reduce(SUM, u64) of `words` table words over the accumulators `who`
// (distinct devices).  Caller holds nothing; this takes g_rccl_mu.
int rccl_sum_tables(qk_accum **who, int n, size_t words) {
  const RcclApi &api = rccl_api();
  if (!api.ok) return fail(QK_ERCCL, "%s", api.why);
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  StdoutToStderr quiet;   // RCCL logs to stdout; the host's stdout is the SVG
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) devs[i] = who[i]->device;
  auto it = g_rccl_comms.find(devs);
  if (it == g_rccl_comms.end()) {
    std::vector<ncclComm_t> comms(n);
    const int e = api.CommInitAll(comms.data(), n, devs.data());
    if (e) return fail(QK_ERCCL, "ncclCommInitAll: %s", api.GetErrorString(e));
    if (getenv("QUACK_VERBOSE")) {
      int v = 0;
      if (api.GetVersion) (void)api.GetVersion(&v);
      fprintf(stderr, "[quack] RCCL %d (built against %d): %d communicators, devices", v, NCCL_VERSION_CODE, n);
      for (int i = 0; i < n; ++i) fprintf(stderr, " %d", devs[i]);
      fprintf(stderr, "; all-reduce(SUM, u64) of %zu words\n", words);
    }
    it = g_rccl_comms.emplace(devs, std::move(comms)).first;
  }
  const std::vector<ncclComm_t> &comms = it->second;
  int e = api.GroupStart();
  if (e) return fail(QK_ERCCL, "ncclGroupStart: %s", api.GetErrorString(e));
  for (int i = 0; i < n && !e; ++i) {
    (void)hipSetDevice(who[i]->device);
    e = api.AllReduce(who[i]->d_table, who[i]->d_table, words, kNcclUint64, kNcclSum, comms[i], who[i]->stream);
  }
  const int e2 = api.GroupEnd();
  if (!e) e = e2;
  // the lock is held until the collective has finished on every device: the next
  // caller's collective (same communicators, other streams) must not overtake it
  hipError_t he = hipSuccess;
  for (int i = 0; i < n; ++i) {
    (void)hipSetDevice(who[i]->device);
    const hipError_t h = hipStreamSynchronize(who[i]->stream);
    if (he == hipSuccess) he = h;
  }
  if (e) return fail(QK_ERCCL, "ncclAllReduce: %s", api.GetErrorString(e));
  if (he != hipSuccess) return fail(QK_EHIP, "waiting for the all-reduce: %s", hipGetErrorString(he));
  return QK_OK;
}
}  // namespace

extern "C" {

const char *qk_last_error(void) { return g_err; }
const char *qk_version(void) { return "quack_hip 0.1 (gfx950)"; }

int qk_device_count(int *count) {
  if (!count) return fail(QK_EINVAL, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(QK_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return QK_OK;
}

int qk_debug_set_mode(int mode) {
  g_ablation_mode = mode;
  return QK_OK;
}

// Launch geometry for a batch shape, without a device (tests/test_planner.py
// checks the invariants the kernels rely on over many shapes).
int qk_debug_plan(uint64_t n_reads, uint32_t max_len, int ragged, int adapters, uint32_t bucket_log2,
                  int gapped, int aligned, int n_cu, uint64_t *out /* [16] */) {
  if (!out) return fail(QK_EINVAL, "out is NULL");
  qk_accum a;
  a.n_cu = n_cu > 0 ? n_cu : 256;
  a.adapters = adapters != 0;
  a.bucket_log2 = adapters ? bucket_log2 : 0;
  a.threads = tune_int("threads", a.threads);
  a.unroll = tune_int("unroll", a.unroll);
  a.pipe = tune_int("pipe", a.pipe);
  a.tile = tune_int("tile", a.tile);
  a.wgs_per_cu = tune_int("wgs_per_cu", a.wgs_per_cu);
  Plan pl;
  int rc = make_plan(&a, n_reads, max_len, ragged != 0, gapped != 0, aligned != 0, &pl);
  if (rc) return rc;
  out[0] = pl.n_tiles; out[1] = pl.tile_pos; out[2] = pl.ch; out[3] = pl.rw;
  out[4] = (uint64_t)pl.unroll; out[5] = (uint64_t)pl.pipe; out[6] = pl.reads_per_slice; out[7] = pl.n_slices;
  out[8] = pl.n_blocks; out[9] = qk::hist_lds_bytes(pl.ch, pl.replicas, pl.fused_adapters, pl.bucket_log2, ragged != 0, pl.stage_reads, pl.w16, pl.fh_words);
  out[10] = pl.halo; out[11] = pl.fused_adapters; out[12] = pl.dynamic; out[13] = (pl.aligned ? 1u : 0u) | (pl.w16 ? 2u : 0u);
  out[14] = pl.replicas; out[15] = qk::hist_row_dwords(pl.ch, pl.replicas);
  return QK_OK;
}

// How a fixed-length batch with the adapter scan would be cut into rows (choose_group + the plan of a row), without a device.
int qk_debug_group(uint64_t n_reads, uint32_t read_len, uint32_t stride, uint32_t bucket_log2, uint64_t *out /* [8] */) {
  if (!out) return fail(QK_EINVAL, "out is NULL");
  qk_accum a;
  a.adapters = true;
  a.bucket_log2 = bucket_log2;
  uint32_t group = 1, row_cap = 0;
  Plan pl;
  for (;;) {
    group = choose_group(&a, n_reads, read_len, stride, true, row_cap);
    const uint32_t row = (group - 1u) * stride + read_len;
    int rc = make_plan(&a, n_reads / group, row, false, false, false, &pl, true, false, group * stride);
    if (rc) return rc;
    if (group <= 1 || (pl.w16 && pl.n_tiles == 1 && (uint64_t)pl.rw * (uint32_t)pl.unroll * group <= qk::kFhRing / 2u && (pl.bucket_log2 || !a.bucket_log2)))
      break;
    row_cap = row - 1u;
  }
  out[0] = group; out[1] = (group - 1u) * stride + read_len; out[2] = pl.tile_pos; out[3] = pl.rw; out[4] = (uint64_t)pl.unroll;
  out[5] = pl.w16; out[6] = pl.bucket_log2; out[7] = qk::hist_lds_bytes(pl.ch, pl.replicas, pl.fused_adapters, pl.bucket_log2, false, pl.stage_reads, pl.w16, pl.fh_words);
  return QK_OK;
}

int qk_debug_wide(uint32_t chunks, uint32_t replicas, uint32_t bucket_log2, uint32_t fh_words, uint64_t *out) {
  if (!out || !chunks || !replicas) return fail(QK_EINVAL, "bad argument");
  const qk::WidePlan w = qk::wide_plan(chunks, replicas, bucket_log2, fh_words);
  out[0] = w.planes; out[1] = w.bucket_off; out[2] = w.ring_off; out[3] = w.bytes;
  out[4] = qk::wide_spare_ring(4u * replicas * chunks);
  out[5] = qk::kCandWords16 + 6u * 8u * chunks + 4u;
  out[6] = out[7] = 0;
  return QK_OK;
}

int qk_accum_create(qk_accum **out, int device, const uint32_t *kmer_bitset,
                    uint64_t max_len_hint) {
  if (!out) return fail(QK_EINVAL, "out is NULL");
  *out = nullptr;
  int n = 0;
  Lap lap;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(QK_ENODEV, "no HIP device available (the accumulation path has no CPU fallback)");
  if (device < 0 || device >= n) return fail(QK_EINVAL, "device %d out of range (0..%d)", device, n - 1);
  qk_accum *a = new (std::nothrow) qk_accum();
  if (!a) return fail(QK_ENOMEM, "out of memory");
  a->device = device;
  lap.mark("runtime");
  int rc = QK_OK;
  do {
    if ((rc = set_device(a))) break;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
      rc = fail(QK_EHIP, "hipGetDeviceProperties failed");
      break;
    }
    a->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    a->threads = tune_int("threads", a->threads);
    a->unroll = tune_int("unroll", a->unroll);
    a->pipe = tune_int("pipe", a->pipe);
    a->tile = tune_int("tile", a->tile);
    a->wgs_per_cu = tune_int("wgs_per_cu", a->wgs_per_cu);
    lap.mark("device");
    if (hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&a->side_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&a->set_free[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&a->set_free[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&a->order_ev, hipEventDisableTiming) != hipSuccess) {
      rc = fail(QK_EHIP, "hipStreamCreate / hipEventCreate failed");
      break;
    }
    lap.mark("stream");
    if (kmer_bitset) {
      a->adapters = true;
      rc = qk::upload_kmer_tables(kmer_bitset, &a->d_kmer_bits, &a->d_kmer_filter, &a->filter_bits);
      if (rc) {
        rc = fail(QK_EHIP, "uploading adapter tables failed: %s", hipGetErrorString((hipError_t)rc));
        break;
      }
      std::vector<uint16_t> buckets;
      if (qk::build_kmer_buckets(kmer_bitset, /*max_log2=*/10, &buckets, &a->bucket_log2, &a->bucket_mul)) {
        if (hipMalloc((void **)&a->d_kmer_buckets, buckets.size() * 2) != hipSuccess ||
            hipMemcpy(a->d_kmer_buckets, buckets.data(), buckets.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
          rc = fail(QK_EHIP, "uploading adapter buckets failed");
          break;
        }
      } else {
        a->bucket_log2 = 0;   // huge adapter set: the candidates consult the global bitset
      }
    }
    if (hipMalloc((void **)&a->d_status, sizeof(uint32_t)) != hipSuccess ||
        hipMemset(a->d_status, 0, sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&a->d_queues, (size_t)kQueueRing * kQueueTiles * sizeof(uint32_t)) != hipSuccess) {
      rc = fail(QK_EHIP, "hipMalloc failed");
      break;
    }
    lap.mark("tables");
    if ((rc = grow_table(a, std::max<uint64_t>(max_len_hint, 64)))) break;
    lap.mark("counters");
  } while (0);
  lap.print("qk_accum_create");
  if (rc) {
    qk_accum_destroy(a);
    return rc;
  }
  *out = a;
  return QK_OK;
}

void qk_accum_destroy(qk_accum *a) {
  if (!a) return;
  (void)hipSetDevice(a->device);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 2; ++i) {
    Slot &s = a->slot[i];
    if (s.h_seq) (void)hipHostFree(s.h_seq);
    if (s.h_qual) (void)hipHostFree(s.h_qual);
    if (s.h_off) (void)hipHostFree(s.h_off);
    if (s.h_len) (void)hipHostFree(s.h_len);
    if (s.d_len) (void)hipFree(s.d_len);
    if (s.d_seq) (void)hipFree(s.d_seq);
    if (s.d_qual) (void)hipFree(s.d_qual);
    if (s.d_off) (void)hipFree(s.d_off);
    if (s.d_hit) (void)hipFree(s.d_hit);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  for (auto &tl : a->timed) {
    (void)hipEventDestroy(tl.t0);
    (void)hipEventDestroy(tl.t1);
    if (tl.b0 != tl.t0) (void)hipEventDestroy(tl.b0);
    if (tl.b1 != tl.t1) (void)hipEventDestroy(tl.b1);
  }
  for (auto e : a->event_pool) (void)hipEventDestroy(e);
  if (tune_on("debug_addr"))   // (where this accumulator's buffers lay: tools/ab_inproc.py --addresses)
    fprintf(stderr, "quack_hip: acc table %p order %p %p reach %p %p queues %p stream %p side %p\n", (void *)a->d_table, (void *)a->d_order[0],
            (void *)a->d_order[1], (void *)a->d_reach[0], (void *)a->d_reach[1], (void *)a->d_queues, (void *)a->stream, (void *)a->side);
  if (a->d_queues) (void)hipFree(a->d_queues);
  if (a->d_status) (void)hipFree(a->d_status);
  for (int i = 0; i < kSets; ++i) {
    if (a->d_order[i]) (void)hipFree(a->d_order[i]);
    if (a->d_reach[i]) (void)hipFree(a->d_reach[i]);
    if (a->set_free[i]) (void)hipEventDestroy(a->set_free[i]);
  }
  if (a->side_done) (void)hipEventDestroy(a->side_done);
  if (a->side) (void)hipStreamDestroy(a->side);
  if (a->d_hit_scratch) (void)hipFree(a->d_hit_scratch);
  if (a->d_starts_scratch) (void)hipFree(a->d_starts_scratch);
  if (a->d_kmer_bits) (void)hipFree(a->d_kmer_bits);
  if (a->d_kmer_filter) (void)hipFree(a->d_kmer_filter);
  if (a->d_kmer_buckets) (void)hipFree(a->d_kmer_buckets);
  if (a->d_table) (void)hipFree(a->d_table);
  if (a->d_table32) (void)hipFree(a->d_table32);
  if (a->order_ev) (void)hipEventDestroy(a->order_ev);
  if (a->stream) (void)hipStreamDestroy(a->stream);
  delete a;
}

int qk_accum_configure(qk_accum *a, int threads, int unroll, int tile, int wgs_per_cu) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  if (threads > 0) a->threads = threads;
  if (unroll > 0) a->unroll = unroll;
  if (tile > 0) a->tile = tile;
  if (wgs_per_cu > 0) a->wgs_per_cu = wgs_per_cu;
  return QK_OK;
}

int qk_accum_acquire(qk_accum *a, uint8_t **seq, uint8_t **qual, uint64_t **offsets,
                     uint64_t *cap_bytes, uint64_t *cap_reads) {
  if (!a || !seq || !qual || !offsets) return fail(QK_EINVAL, "NULL argument");
  if (a->held_slot >= 0) return fail(QK_ESTATE, "a batch is already acquired");
  int rc = set_device(a);
  if (rc) return rc;
  if ((rc = ensure_slot(a, a->next_slot))) return rc;
  Slot &s = a->slot[a->next_slot];
  if (s.busy) {
    QK_HIP(hipEventSynchronize(s.done));
    s.busy = false;
  }
  a->held_slot = a->next_slot;
  *seq = s.h_seq;
  *qual = s.h_qual;
  *offsets = s.h_off;
  if (cap_bytes) *cap_bytes = a->cap_bytes;
  if (cap_reads) *cap_reads = a->cap_reads;
  return QK_OK;
}

int qk_accum_resize_slots(qk_accum *a, uint64_t min_bytes) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  if (a->held_slot >= 0) return fail(QK_ESTATE, "a batch is acquired");
  if (min_bytes > (1ull << 40)) return fail(QK_EINVAL, "slot size out of range");
  if (a->cap_bytes && min_bytes <= a->cap_bytes) return QK_OK;
  int rc = set_device(a);
  if (rc) return rc;
  // rare (a read longer than every one before it by far): wait for both slots, drop them,
  // and let the next acquire allocate at the new size
  for (int i = 0; i < 2; ++i) {
    Slot &s = a->slot[i];
    if (s.busy) {
      QK_HIP(hipEventSynchronize(s.done));
      s.busy = false;
    }
    if (s.stream) QK_HIP(hipStreamSynchronize(s.stream));
    if (s.stream && a->order_stream == s.stream) a->order_valid = false;   // (all of it has finished; the stream goes away)
    if (s.h_seq) (void)hipHostFree(s.h_seq);
    if (s.h_qual) (void)hipHostFree(s.h_qual);
    if (s.h_off) (void)hipHostFree(s.h_off);
    if (s.h_len) (void)hipHostFree(s.h_len);
    if (s.d_len) (void)hipFree(s.d_len);
    if (s.d_seq) (void)hipFree(s.d_seq);
    if (s.d_qual) (void)hipFree(s.d_qual);
    if (s.d_off) (void)hipFree(s.d_off);
    if (s.d_hit) (void)hipFree(s.d_hit);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.done) (void)hipEventDestroy(s.done);
    s = Slot{};
  }
  a->cap_bytes = round_up(min_bytes, 1u << 20);
  a->cap_reads = a->cap_bytes / 32 + 1024;
  return QK_OK;
}

int qk_accum_commit(qk_accum *a, uint64_t n_reads, uint64_t total, int offsets_used,
                    uint32_t read_len) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  if (a->held_slot < 0) return fail(QK_ESTATE, "no batch acquired");
  Slot &s = a->slot[a->held_slot];
  if (total > a->cap_bytes || n_reads > a->cap_reads)
    return fail(QK_EINVAL, "batch exceeds slot capacity");
  int rc = set_device(a);
  if (rc) return rc;
  uint32_t max_len = read_len;
  if (offsets_used) {
    if (s.h_off[0] != 0 || s.h_off[n_reads] != total)
      return fail(QK_EINVAL, "offsets[0] must be 0 and offsets[n] the byte total");
    uint64_t m = 0;
    for (uint64_t i = 0; i < n_reads; ++i) {
      if (s.h_off[i + 1] < s.h_off[i]) return fail(QK_EINVAL, "offsets not monotonic at read %llu", (unsigned long long)i);
      m = std::max(m, s.h_off[i + 1] - s.h_off[i]);
    }
    if (m > 0xFFFFFFF0ull) return fail(QK_EINVAL, "read too long");
    max_len = (uint32_t)m;
  } else if ((uint64_t)read_len * n_reads != total) {
    return fail(QK_EINVAL, "fixed batch: n_reads*read_len != total");
  }
  a->held_slot = -1;
  a->next_slot ^= 1;
  if (n_reads == 0) return QK_OK;
  // grow before enqueueing copies so that the (synchronising) growth cannot
  // race with this slot's stream
  if ((rc = grow_table(a, std::max<uint64_t>(max_len, 11)))) return rc;
  // pad the tail so that the 8-byte chunk loads past the last read are defined
  memset(s.h_seq + total, 0, QK_TAIL_SLACK);
  memset(s.h_qual + total, 0, QK_TAIL_SLACK);
  QK_HIP(hipMemcpyAsync(s.d_seq, s.h_seq, total + QK_TAIL_SLACK, hipMemcpyHostToDevice, s.stream));
  QK_HIP(hipMemcpyAsync(s.d_qual, s.h_qual, total + QK_TAIL_SLACK, hipMemcpyHostToDevice, s.stream));
  if (offsets_used)
    QK_HIP(hipMemcpyAsync(s.d_off, s.h_off, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s.stream));
  rc = enqueue_batch(a, s.d_seq, s.d_qual, offsets_used ? s.d_off : nullptr, s.d_hit,
                     n_reads, total, max_len, s.stream);
  if (rc) return rc;
  QK_HIP(hipEventRecord(s.done, s.stream));
  s.busy = true;
  return QK_OK;
}

int qk_accum_padded_stride(qk_accum *a, uint32_t read_len, uint32_t *stride) {
  if (!a || !stride) return fail(QK_EINVAL, "NULL argument");
  *stride = padded_stride_for(a, read_len);
  return QK_OK;
}

int qk_accum_commit_padded(qk_accum *a, uint64_t n_reads, uint32_t read_len, uint32_t stride) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  if (a->held_slot < 0) return fail(QK_ESTATE, "no batch acquired");
  if (stride == 0 || (stride & 3u) || read_len > stride) return fail(QK_EINVAL, "stride must be a multiple of 4 and >= read_len");
  if (read_len == stride) return qk_accum_commit(a, n_reads, n_reads * (uint64_t)stride, 0, read_len);
  Slot &s = a->slot[a->held_slot];
  const uint64_t total = n_reads * (uint64_t)stride;
  if (total > a->cap_bytes || n_reads > a->cap_reads) return fail(QK_EINVAL, "batch exceeds slot capacity");
  int rc = set_device(a);
  if (rc) return rc;
  a->held_slot = -1;
  a->next_slot ^= 1;
  if (n_reads == 0) return QK_OK;
  if (read_len == 0) {   // only empty reads: they count as sequences, nothing else
    a->n_reads += n_reads;
    return QK_OK;
  }
  if ((rc = grow_table(a, std::max<uint64_t>(read_len, 11)))) return rc;
  memset(s.h_seq + total, 0, QK_TAIL_SLACK);
  memset(s.h_qual + total, 0, QK_TAIL_SLACK);
  QK_HIP(hipMemcpyAsync(s.d_seq, s.h_seq, total + QK_TAIL_SLACK, hipMemcpyHostToDevice, s.stream));
  QK_HIP(hipMemcpyAsync(s.d_qual, s.h_qual, total + QK_TAIL_SLACK, hipMemcpyHostToDevice, s.stream));
  rc = enqueue_batch(a, s.d_seq, s.d_qual, nullptr, s.d_hit, n_reads, total, read_len, s.stream, nullptr, 0, stride);
  if (rc) return rc;
  QK_HIP(hipEventRecord(s.done, s.stream));
  s.busy = true;
  return QK_OK;
}

int qk_accum_submit(qk_accum *a, const uint8_t *seq, const uint8_t *qual,
                    const uint64_t *offsets, uint64_t n_reads) {
  if (!a || !offsets || (n_reads && (!seq || !qual))) return fail(QK_EINVAL, "NULL argument");
  for (uint64_t k = 0; k < n_reads; ++k)
    if (offsets[k + 1] < offsets[k]) return fail(QK_EINVAL, "offsets not monotonic at read %llu", (unsigned long long)k);
  uint64_t i = 0;
  while (i < n_reads) {
    uint8_t *hs, *hq;
    uint64_t *ho, capb, capr;
    int rc = qk_accum_acquire(a, &hs, &hq, &ho, &capb, &capr);
    if (rc) return rc;
    const uint64_t base = offsets[i];
    uint64_t j = i;
    while (j < n_reads && j - i < capr && offsets[j + 1] - base <= capb) ++j;
    if (j == i) {
      // one read longer than a slot: bigger slots (a batch holds whole reads)
      a->held_slot = -1;
      if ((rc = qk_accum_resize_slots(a, offsets[i + 1] - base))) return rc;
      continue;
    }
    const uint64_t bytes = offsets[j] - base;
    memcpy(hs, seq + base, bytes);
    memcpy(hq, qual + base, bytes);
    for (uint64_t k = i; k <= j; ++k) ho[k - i] = offsets[k] - base;
    if ((rc = qk_accum_commit(a, j - i, bytes, 1, 0))) {
      a->held_slot = -1;   // the copying feed owns the slot: give it back on failure
      return rc;
    }
    i = j;
  }
  return QK_OK;
}

int qk_accum_submit_fixed(qk_accum *a, const uint8_t *seq, const uint8_t *qual,
                          uint32_t read_len, uint64_t n_reads) {
  if (!a || (n_reads && read_len && (!seq || !qual))) return fail(QK_EINVAL, "NULL argument");
  if (read_len == 0) {
    a->n_reads += n_reads;
    return QK_OK;
  }
  // the copy into the pinned slot re-lays the reads at a padded stride when that gives them the dword-aligned kernels
  const uint32_t stride = padded_stride_for(a, read_len);
  const uint32_t step = stride ? stride : read_len;
  uint64_t i = 0;
  while (i < n_reads) {
    uint8_t *hs, *hq;
    uint64_t *ho, capb, capr;
    int rc = qk_accum_acquire(a, &hs, &hq, &ho, &capb, &capr);
    if (rc) return rc;
    uint64_t n = std::min<uint64_t>(n_reads - i, capb / step);
    if (n == 0) {
      a->held_slot = -1;
      if ((rc = qk_accum_resize_slots(a, step))) return rc;
      continue;
    }
    if (stride) {
      for (uint64_t r = 0; r < n; ++r) {
        memcpy(hs + r * stride, seq + (i + r) * read_len, read_len);
        memcpy(hq + r * stride, qual + (i + r) * read_len, read_len);
        memset(hs + r * stride + read_len, 0, stride - read_len);   // (never counted; kept defined)
        memset(hq + r * stride + read_len, 0, stride - read_len);
      }
      rc = qk_accum_commit_padded(a, n, read_len, stride);
    } else {
      memcpy(hs, seq + i * read_len, n * read_len);
      memcpy(hq, qual + i * read_len, n * read_len);
      rc = qk_accum_commit(a, n, n * read_len, 0, read_len);
    }
    if (rc) {
      a->held_slot = -1;
      return rc;
    }
    i += n;
  }
  return QK_OK;
}

int qk_accum_submit_device(qk_accum *a, const void *d_seq, const void *d_qual,
                           const void *d_offsets, uint64_t n_reads, uint64_t total_bytes,
                           uint32_t max_len, void *hip_stream) {
  if (!a || (n_reads && max_len && (!d_seq || !d_qual))) return fail(QK_EINVAL, "NULL argument");
  int rc = set_device(a);
  if (rc) return rc;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : a->stream;
  if (hip_stream) a->foreign_stream_used = true;
  uint32_t *d_hit = nullptr;
  if (a->adapters) {
    if (a->hit_scratch_reads < n_reads) {
      QK_HIP(hipDeviceSynchronize());
      if (a->d_hit_scratch) QK_HIP(hipFree(a->d_hit_scratch));
      a->d_hit_scratch = nullptr;
      a->hit_scratch_reads = 0;
      QK_HIP(hipMalloc((void **)&a->d_hit_scratch, n_reads * sizeof(uint32_t)));
      a->hit_scratch_reads = n_reads;
    }
    d_hit = a->d_hit_scratch;
  }
  return enqueue_batch(a, (const uint8_t *)d_seq, (const uint8_t *)d_qual,
                       (const uint64_t *)d_offsets, d_hit, n_reads, total_bytes, max_len, st);
}

int qk_accum_submit_device_gapped(qk_accum *a, const void *d_seq, const void *d_qual,
                                  const void *d_starts, const void *d_lengths,
                                  uint64_t n_reads, uint64_t extent_bytes,
                                  uint32_t max_len, uint32_t flags, void *hip_stream) {
  if (!a || (n_reads && (!d_starts || !d_lengths || (max_len && (!d_seq || !d_qual))))) return fail(QK_EINVAL, "NULL argument");
  if (flags & ~QK_BATCH_ALIGNED128) return fail(QK_EINVAL, "unknown flags 0x%x", flags);
  int rc = set_device(a);
  if (rc) return rc;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : a->stream;
  if (hip_stream) a->foreign_stream_used = true;
  uint32_t *d_hit = nullptr;
  if (a->adapters) {
    if (a->hit_scratch_reads < n_reads) {
      QK_HIP(hipDeviceSynchronize());
      if (a->d_hit_scratch) QK_HIP(hipFree(a->d_hit_scratch));
      a->d_hit_scratch = nullptr;
      a->hit_scratch_reads = 0;
      QK_HIP(hipMalloc((void **)&a->d_hit_scratch, n_reads * sizeof(uint32_t)));
      a->hit_scratch_reads = n_reads;
    }
    d_hit = a->d_hit_scratch;
  }
  return enqueue_batch(a, (const uint8_t *)d_seq, (const uint8_t *)d_qual, (const uint64_t *)d_starts, d_hit,
                       n_reads, extent_bytes, max_len, st, (const uint32_t *)d_lengths, flags);
}

int qk_accum_submit_device_strided(qk_accum *a, const void *d_seq, const void *d_qual, const void *d_lengths,
                                   uint64_t n_reads, uint32_t stride, uint32_t max_len, void *hip_stream) {
  return qk_accum_submit_device_strided_flags(a, d_seq, d_qual, d_lengths, n_reads, stride, max_len, 0u, hip_stream);
}

int qk_accum_submit_device_strided_flags(qk_accum *a, const void *d_seq, const void *d_qual, const void *d_lengths,
                                         uint64_t n_reads, uint32_t stride, uint32_t max_len, uint32_t flags, void *hip_stream) {
  if (!a || (n_reads && max_len && (!d_seq || !d_qual))) return fail(QK_EINVAL, "NULL argument");
  if (flags & ~QK_BATCH_NEUTRAL_PADS) return fail(QK_EINVAL, "unknown flags 0x%x", flags);
  if ((flags & QK_BATCH_NEUTRAL_PADS) && !d_lengths) return fail(QK_EINVAL, "QK_BATCH_NEUTRAL_PADS is a promise about a batch with lengths[]");
  if (stride == 0 || (stride & 3u) || max_len > stride) return fail(QK_EINVAL, "stride must be a multiple of 4 and >= max_len");
  // d_lengths == NULL: every read is max_len long — a fixed-length batch at a padded stride
  if ((((uintptr_t)d_seq | (uintptr_t)d_qual) & 3u) != 0) return fail(QK_EINVAL, "strided batches must start on a 4-byte boundary");
  int rc = set_device(a);
  if (rc) return rc;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : a->stream;
  if (hip_stream) a->foreign_stream_used = true;
  uint32_t *d_hit = nullptr;
  if (a->adapters) {
    if (a->hit_scratch_reads < n_reads) {
      QK_HIP(hipDeviceSynchronize());
      if (a->d_hit_scratch) QK_HIP(hipFree(a->d_hit_scratch));
      a->d_hit_scratch = nullptr;
      a->hit_scratch_reads = 0;
      QK_HIP(hipMalloc((void **)&a->d_hit_scratch, n_reads * sizeof(uint32_t)));
      a->hit_scratch_reads = n_reads;
    }
    d_hit = a->d_hit_scratch;
  }
  return enqueue_batch(a, (const uint8_t *)d_seq, (const uint8_t *)d_qual, nullptr, d_hit, n_reads,
                       n_reads * (uint64_t)stride, max_len, st, (const uint32_t *)d_lengths, flags, stride);
}

int qk_accum_submit_strided(qk_accum *a, const uint8_t *seq, const uint8_t *qual, const uint32_t *lengths,
                            uint32_t stride, uint64_t n_reads) {
  if (!a || (n_reads && (!lengths || !seq || !qual))) return fail(QK_EINVAL, "NULL argument");
  if (stride == 0 || (stride & 3u)) return fail(QK_EINVAL, "stride must be a multiple of 4");
  uint64_t i = 0;
  while (i < n_reads) {
    uint8_t *hs, *hq;
    uint64_t *ho, capb, capr;
    uint32_t *hl;
    int rc = qk_accum_acquire(a, &hs, &hq, &ho, &capb, &capr);
    if (rc) return rc;
    if ((rc = qk_accum_slot_lengths(a, &hl))) return rc;
    const uint64_t n = std::min<uint64_t>(std::min<uint64_t>(n_reads - i, capb / stride), capr);
    if (n == 0) {
      a->held_slot = -1;
      return fail(QK_EINVAL, "stride exceeds the batch slot");
    }
    memcpy(hs, seq + i * stride, n * stride);
    memcpy(hq, qual + i * stride, n * stride);
    memcpy(hl, lengths + i, n * sizeof(uint32_t));
    // the slot is ours: the bytes behind every read become 0xFF, and the batch runs the kernel without tail masks
    const bool neutral = !getenv("QUACK_HIP_NO_NEUTRAL");
    for (uint64_t r = 0; neutral && r < n; ++r)
      if (hl[r] < stride) {
        memset(hs + r * stride + hl[r], 0xFF, stride - hl[r]);
        memset(hq + r * stride + hl[r], 0xFF, stride - hl[r]);
      }
    if ((rc = qk_accum_commit_strided_flags(a, n, stride, neutral ? QK_BATCH_NEUTRAL_PADS : 0u))) {
      a->held_slot = -1;
      return rc;
    }
    i += n;
  }
  return QK_OK;
}

int qk_accum_commit_strided(qk_accum *a, uint64_t n_reads, uint32_t stride) {
  return qk_accum_commit_strided_flags(a, n_reads, stride, 0u);
}

int qk_accum_commit_strided_flags(qk_accum *a, uint64_t n_reads, uint32_t stride, uint32_t flags) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  if (flags & ~QK_BATCH_NEUTRAL_PADS) return fail(QK_EINVAL, "unknown flags 0x%x", flags);
  if (a->held_slot < 0) return fail(QK_ESTATE, "no batch acquired");
  if (stride == 0 || (stride & 3u)) return fail(QK_EINVAL, "stride must be a multiple of 4");
  Slot &s = a->slot[a->held_slot];
  const uint64_t total = n_reads * (uint64_t)stride;
  if (total > a->cap_bytes || n_reads > a->cap_reads) return fail(QK_EINVAL, "batch exceeds slot capacity");
  int rc = set_device(a);
  if (rc) return rc;
  uint32_t max_len = 0;
  for (uint64_t i = 0; i < n_reads; ++i) {
    const uint32_t l = s.h_len[i];
    if (l > stride) return fail(QK_EINVAL, "read %llu is longer than the stride", (unsigned long long)i);
    max_len = std::max(max_len, l);
    // the promise is checked where the host has the bytes: every pad byte of every read (a few bytes per read — a pad
    // that is T / C / G would be counted while `valid` comes from the lengths: content[A] would underflow silently)
    if ((flags & QK_BATCH_NEUTRAL_PADS) && l < stride) {
      const uint8_t *ps = s.h_seq + i * (uint64_t)stride, *pq = s.h_qual + i * (uint64_t)stride;
      uint8_t all = 0xFF;
      for (uint32_t k = l; k < stride; ++k) all &= ps[k] & pq[k];
      if (all != 0xFF)
        return fail(QK_EINVAL, "read %llu: QK_BATCH_NEUTRAL_PADS promised 0xFF behind the read's last base", (unsigned long long)i);
    }
  }
  a->held_slot = -1;
  a->next_slot ^= 1;
  if (n_reads == 0) return QK_OK;
  if ((rc = grow_table(a, std::max<uint64_t>(max_len, 11)))) return rc;
  memset(s.h_seq + total, 0, QK_TAIL_SLACK);
  memset(s.h_qual + total, 0, QK_TAIL_SLACK);
  QK_HIP(hipMemcpyAsync(s.d_seq, s.h_seq, total + QK_TAIL_SLACK, hipMemcpyHostToDevice, s.stream));
  QK_HIP(hipMemcpyAsync(s.d_qual, s.h_qual, total + QK_TAIL_SLACK, hipMemcpyHostToDevice, s.stream));
  QK_HIP(hipMemcpyAsync(s.d_len, s.h_len, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream));
  rc = enqueue_batch(a, s.d_seq, s.d_qual, nullptr, s.d_hit, n_reads, total, max_len, s.stream, s.d_len, flags, stride);
  if (rc) return rc;
  QK_HIP(hipEventRecord(s.done, s.stream));
  s.busy = true;
  return QK_OK;
}

int qk_accum_slot_lengths(qk_accum *a, uint32_t **lengths) {
  if (!a || !lengths) return fail(QK_EINVAL, "NULL argument");
  if (a->held_slot < 0) return fail(QK_ESTATE, "no batch acquired");
  *lengths = a->slot[a->held_slot].h_len;
  return QK_OK;
}

int qk_accum_commit_gapped(qk_accum *a, uint64_t n_reads, uint64_t extent, uint32_t flags) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  if (a->held_slot < 0) return fail(QK_ESTATE, "no batch acquired");
  if (flags & ~QK_BATCH_ALIGNED128) return fail(QK_EINVAL, "unknown flags 0x%x", flags);
  Slot &s = a->slot[a->held_slot];
  if (extent > a->cap_bytes || n_reads > a->cap_reads) return fail(QK_EINVAL, "batch exceeds slot capacity");
  int rc = set_device(a);
  if (rc) return rc;
  uint64_t end = 0;
  uint32_t max_len = 0;
  for (uint64_t i = 0; i < n_reads; ++i) {
    if (s.h_off[i] < end) return fail(QK_EINVAL, "read %llu starts inside its predecessor", (unsigned long long)i);
    if ((flags & QK_BATCH_ALIGNED128) && (s.h_off[i] & 127u)) return fail(QK_EINVAL, "read %llu is not 128-byte aligned", (unsigned long long)i);
    end = s.h_off[i] + s.h_len[i];
    max_len = std::max(max_len, s.h_len[i]);
  }
  if (end > extent) return fail(QK_EINVAL, "reads end at %llu, past the batch extent", (unsigned long long)end);
  a->held_slot = -1;
  a->next_slot ^= 1;
  if (n_reads == 0) return QK_OK;
  if ((rc = grow_table(a, std::max<uint64_t>(max_len, 11)))) return rc;
  memset(s.h_seq + extent, 0, QK_TAIL_SLACK);
  memset(s.h_qual + extent, 0, QK_TAIL_SLACK);
  QK_HIP(hipMemcpyAsync(s.d_seq, s.h_seq, extent + QK_TAIL_SLACK, hipMemcpyHostToDevice, s.stream));
  QK_HIP(hipMemcpyAsync(s.d_qual, s.h_qual, extent + QK_TAIL_SLACK, hipMemcpyHostToDevice, s.stream));
  QK_HIP(hipMemcpyAsync(s.d_off, s.h_off, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, s.stream));
  QK_HIP(hipMemcpyAsync(s.d_len, s.h_len, n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream));
  rc = enqueue_batch(a, s.d_seq, s.d_qual, s.d_off, s.d_hit, n_reads, extent, max_len, s.stream, s.d_len, flags);
  if (rc) return rc;
  QK_HIP(hipEventRecord(s.done, s.stream));
  s.busy = true;
  return QK_OK;
}

int qk_accum_sync(qk_accum *a) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  int rc = set_device(a);
  if (rc) return rc;
  if (a->foreign_stream_used) {
    QK_HIP(hipDeviceSynchronize());
  } else {
    QK_HIP(hipStreamSynchronize(a->stream));
    if (a->side) QK_HIP(hipStreamSynchronize(a->side));
    for (int i = 0; i < 2; ++i)
      if (a->slot[i].stream) QK_HIP(hipStreamSynchronize(a->slot[i].stream));
  }
  for (int i = 0; i < 2; ++i) a->slot[i].busy = false;
  if (a->status_armed) {
    uint32_t st = 0;
    QK_HIP(hipMemcpy(&st, a->d_status, sizeof st, hipMemcpyDeviceToHost));
    a->status_armed = false;
    if (st) {
      QK_HIP(hipMemset(a->d_status, 0, sizeof st));
      if (st & qk::kStatusBadLength)
        return fail(QK_EINVAL, "a strided batch holds a read longer than its stride; the counters are unusable");
      if (st & qk::kStatusBadPads)
        return fail(QK_EINVAL, "a batch submitted as QK_BATCH_NEUTRAL_PADS holds a pad byte that is not 0xFF; the counters are unusable");
      return fail(QK_EINVAL, "a batch submitted as QK_BATCH_ALIGNED128 holds a read that does not start on a 128-byte boundary; the counters are unusable");
    }
  }
  return drain_timing(a);
}

int qk_accum_stats(qk_accum *a, uint64_t *max_len, uint64_t *n_reads) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  if (max_len) *max_len = a->max_len;
  if (n_reads) *n_reads = a->n_reads;
  return QK_OK;
}

// everything submitted has run AND the 32-bit side table is part of d_table: what every reader of the table calls first
static int settle(qk_accum *a) {
  int rc = qk_accum_sync(a);
  if (rc) return rc;
  if (a->reads32) {
    if ((rc = fold_table32(a, a->stream))) return rc;
    QK_HIP(hipStreamSynchronize(a->stream));
  }
  return QK_OK;
}

int qk_accum_table_words(qk_accum *a, uint64_t *n_words) {
  if (!a || !n_words) return fail(QK_EINVAL, "NULL argument");
  *n_words = (uint64_t)QK_N_ROWS * a->table_len + 1;
  return QK_OK;
}

int qk_accum_reserve(qk_accum *a, uint64_t max_len) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  int rc = set_device(a);
  if (rc) return rc;
  return grow_table(a, max_len, /*exact=*/true);
}

int qk_accum_export_table(qk_accum *a, void *d_dst, void *hip_stream) {
  if (!a || !d_dst) return fail(QK_EINVAL, "NULL argument");
  int rc = settle(a);
  if (rc) return rc;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : a->stream;
  const size_t words = (size_t)QK_N_ROWS * a->table_len;
  unsigned long long n = a->n_reads;
  QK_HIP(hipMemcpyAsync(a->d_table + words, &n, 8, hipMemcpyHostToDevice, st));
  QK_HIP(hipMemcpyAsync(d_dst, a->d_table, (words + 1) * 8, hipMemcpyDeviceToDevice, st));
  QK_HIP(hipStreamSynchronize(st));
  return QK_OK;
}

int qk_accum_import_table(qk_accum *a, const void *d_src, uint64_t max_len, void *hip_stream) {
  if (!a || !d_src) return fail(QK_EINVAL, "NULL argument");
  int rc = settle(a);
  if (rc) return rc;
  if (max_len > a->table_len) return fail(QK_EINVAL, "max_len exceeds the table; call qk_accum_reserve first");
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : a->stream;
  const size_t words = (size_t)QK_N_ROWS * a->table_len;
  unsigned long long n = 0;
  QK_HIP(hipMemcpyAsync(a->d_table, d_src, (words + 1) * 8, hipMemcpyDeviceToDevice, st));
  QK_HIP(hipMemcpyAsync(&n, a->d_table + words, 8, hipMemcpyDeviceToHost, st));
  QK_HIP(hipStreamSynchronize(st));
  a->n_reads = n;
  a->max_len = max_len;
  return QK_OK;
}

int qk_accum_allreduce(qk_accum **accs, int n) {
  if (!accs || n <= 0) return fail(QK_EINVAL, "bad arguments");
  for (int i = 0; i < n; ++i)
    if (!accs[i]) return fail(QK_EINVAL, "accs[%d] is NULL", i);
  // common geometry: the longest read seen by any shard (ncclMax on one word
  // in a multi-process job; here the host already knows every shard)
  uint64_t max_len = 0, table_len = 0, total_reads = 0;
  for (int i = 0; i < n; ++i) {
    int rc = settle(accs[i]);
    if (rc) return rc;
    max_len = std::max(max_len, accs[i]->max_len);
    table_len = std::max(table_len, accs[i]->table_len);
    total_reads += accs[i]->n_reads;
  }
  for (int i = 0; i < n; ++i) {
    int rc = set_device(accs[i]);
    if (rc) return rc;
    if ((rc = grow_table(accs[i], table_len, /*exact=*/true))) return rc;
    if (accs[i]->table_len != table_len) return fail(QK_ESTATE, "table sizes diverged");
  }
  const size_t words = (size_t)QK_N_ROWS * table_len;
  // 1) accumulators that share a device are summed into that device's first
  //    one (the "leader") with a plain add kernel — no collective needed
  std::vector<int> leader(n);
  std::vector<int> leaders;
  for (int i = 0; i < n; ++i) {
    leader[i] = i;
    for (int j = 0; j < i; ++j)
      if (accs[j]->device == accs[i]->device) {
        leader[i] = leader[j];
        break;
      }
    if (leader[i] == i) leaders.push_back(i);
  }
  for (int i = 0; i < n; ++i) {
    if (leader[i] == i) continue;
    qk_accum *dst = accs[leader[i]];
    int rc = set_device(dst);
    if (rc) return rc;
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 4096);
    hipLaunchKernelGGL(qk::table_add_kernel, dim3(blocks), dim3(256), 0, dst->stream, dst->d_table, accs[i]->d_table, words);
    QK_HIP(hipGetLastError());
  }
  for (int l : leaders) {
    int rc = set_device(accs[l]);
    if (rc) return rc;
    QK_HIP(hipStreamSynchronize(accs[l]->stream));
  }
  // 2) distinct devices: ONE all-reduce of the integer tables over xGMI
  //    (QUACK_HIP_RCCL_ALWAYS=1: also with a single device — lets a one-GPU box
  //    exercise the RCCL call sequence, tests/test_gpu_multi.py)
  const int nl = (int)leaders.size();
  if (nl > 1 || getenv("QUACK_HIP_RCCL_ALWAYS")) {
    std::vector<qk_accum *> who(nl);
    for (int i = 0; i < nl; ++i) who[i] = accs[leaders[i]];
    const int rc = rccl_sum_tables(who.data(), nl, words);
    if (rc) return rc;
  }
  // 3) every accumulator ends up holding the global table
  for (int i = 0; i < n; ++i) {
    if (leader[i] != i) {
      int rc = set_device(accs[i]);
      if (rc) return rc;
      QK_HIP(hipMemcpy(accs[i]->d_table, accs[leader[i]]->d_table, words * 8, hipMemcpyDeviceToDevice));
    }
    accs[i]->max_len = max_len;
    accs[i]->n_reads = total_reads;
  }
  return QK_OK;
}

int qk_accum_finish(qk_accum *a, qk_base_info *out, uint64_t cap_positions,
                    uint64_t *max_len, uint64_t *n_reads) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  int rc = settle(a);
  if (rc) return rc;
#ifdef QK_TIMING   /* experiment: the phases of the last histogram launch, per workgroup (qk::qk_timing) */
  if (out) {
    static unsigned long long h[1024 * 16];
    QK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(qk::qk_timing), sizeof h));
    unsigned long long t_first = ~0ull, t_last = 0;
    int nb = 0;
    for (int b = 0; b < 1024; ++b)
      if (h[b * 16]) {
        ++nb;
        t_first = std::min(t_first, h[b * 16]);
        t_last = std::max(t_last, h[b * 16 + 6]);
      }
    fprintf(stderr, "[timing] %d workgroups, first start -> last end %.2f us\n", nb, (double)(t_last - t_first) / 100.0);
    const char *names[6] = {"tables + LDS clear", "ring reset + prologue", "step loop", "fold", "spill", "flush"};
    double start_mean = 0, start_max = 0;
    for (int b = 0; b < 1024; ++b)
      if (h[b * 16]) {
        const double d = (double)(h[b * 16] - t_first) / 100.0;
        start_mean += d / nb;
        start_max = std::max(start_max, d);
      }
    fprintf(stderr, "[timing]   start after the first workgroup's: mean %.2f max %.2f us\n", start_mean, start_max);
    {   // when the workgroups end, and whether that goes with the XCD (workgroup b runs on XCD b % 8)
      double end_mean = 0, end_min = 1e30, end_max = 0, by_xcd[8] = {0};
      int n_xcd[8] = {0};
      for (int b = 0; b < 1024; ++b)
        if (h[b * 16]) {
          const double d = (double)(h[b * 16 + 6] - t_first) / 100.0;
          end_mean += d / nb;
          end_min = std::min(end_min, d);
          end_max = std::max(end_max, d);
          by_xcd[b & 7] += d;
          n_xcd[b & 7]++;
        }
      fprintf(stderr, "[timing]   workgroups end (after the first start): mean %.2f min %.2f max %.2f us; by XCD:", end_mean, end_min, end_max);
      for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", n_xcd[x] ? by_xcd[x] / n_xcd[x] : 0.0);
      fprintf(stderr, "; by eighth of the grid:");
      for (int o = 0; o < 8; ++o) {
        double m = 0;
        int c = 0;
        for (int b = o * nb / 8; b < (o + 1) * nb / 8; ++b)
          if (h[b * 16]) m += (double)(h[b * 16 + 6] - t_first) / 100.0, ++c;
        fprintf(stderr, " %.1f", c ? m / c : 0.0);
      }
      fprintf(stderr, "\n");
    }
    {
      static unsigned long long wt[1024 * 16];
      QK_HIP(hipMemcpyFromSymbol(wt, HIP_SYMBOL(qk::qk_wtime), sizeof wt));
      double skew = 0, skew_max = 0, tail = 0, w0 = 0;
      for (int b = 0; b < 1024; ++b)
        if (h[b * 16]) {
          unsigned long long lo = ~0ull, hi = 0;
          for (int w = 0; w < 16; ++w) lo = std::min(lo, wt[b * 16 + w]), hi = std::max(hi, wt[b * 16 + w]);
          skew += (double)(hi - lo) / 100.0 / nb;
          skew_max = std::max(skew_max, (double)(hi - lo) / 100.0);
          tail += (double)((long long)(h[b * 16 + 6] - hi)) / 100.0 / nb;
          w0 += (double)((long long)(wt[b * 16] - lo)) / 100.0 / nb;
        }
      {
        double per[16] = {0};
        for (int b = 0; b < 1024; ++b)
          if (h[b * 16]) {
            unsigned long long lo = ~0ull;
            for (int w = 0; w < 16; ++w) lo = std::min(lo, wt[b * 16 + w]);
            for (int w = 0; w < 16; ++w) per[w] += (double)(wt[b * 16 + w] - lo) / 100.0 / nb;
          }
        fprintf(stderr, "[timing]   wave w leaves the loop this long after the workgroup's first (mean, us):");
        for (int w = 0; w < 16; ++w) fprintf(stderr, " %.1f", per[w]);
        fprintf(stderr, "\n");
      }
      fprintf(stderr, "[timing]   waves of a workgroup leave the step loop %.2f us apart (mean; max %.2f; wave 0 %.2f after the first); "
              "last wave out -> workgroup done: %.2f us\n", skew, skew_max, w0, tail);
    }
    {
      const char *fn[4] = {"flush: wait for the waves", "flush: quality rows", "flush: content rows", "flush: the rest + atomics done"};
      const int a[4] = {5, 8, 9, 10}, e[4] = {8, 9, 10, 6};
      for (int i = 0; i < 4; ++i) {
        double mean = 0;
        for (int b = 0; b < 1024; ++b)
          if (h[b * 16]) mean += (double)((long long)(h[b * 16 + e[i]] - h[b * 16 + a[i]])) / 100.0 / nb;
        fprintf(stderr, "[timing]   %-32s mean %.2f us\n", fn[i], mean);
      }
    }
    for (int i = 0; i < 6; ++i) {
      double mean = 0, mx = 0, mn = 1e30;
      for (int b = 0; b < 1024; ++b)
        if (h[b * 16]) {
          const double d = (double)((long long)(h[b * 16 + i + 1] - h[b * 16 + i])) / 100.0;
          mean += d / nb;
          mx = std::max(mx, d);
          mn = std::min(mn, d);
        }
      fprintf(stderr, "[timing]   %-22s mean %.2f min %.2f max %.2f us\n", names[i], mean, mn, mx);
    }
  }
#endif
  if (max_len) *max_len = a->max_len;
  if (n_reads) *n_reads = a->n_reads;
  if (!out) return QK_OK;
  if (cap_positions < a->max_len) return fail(QK_EINVAL, "output holds %llu positions, need %llu",
                                               (unsigned long long)cap_positions, (unsigned long long)a->max_len);
  if (a->max_len == 0) return QK_OK;
  const uint64_t ml = a->max_len;
  std::vector<uint64_t> planar((size_t)QK_N_ROWS * ml);
  QK_HIP(hipMemcpy2D(planar.data(), ml * 8, a->d_table, a->table_len * 8, ml * 8, QK_N_ROWS,
                     hipMemcpyDeviceToHost));
  uint64_t *dst = reinterpret_cast<uint64_t *>(out);
  for (uint64_t pos = 0; pos < ml; ++pos)
    for (int row = 0; row < QK_N_ROWS; ++row)
      dst[pos * QK_N_ROWS + row] = planar[(size_t)row * ml + pos];
  return QK_OK;
}

int qk_accum_timing_enable(qk_accum *a, int on) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  a->timing = on < 0 ? 0 : on;
  a->timing_seq = 0;
  a->timing_ms = 0;
  a->timing_batch_ms = 0;
  a->timing_min_ms = a->timing_max_ms = 0;
  a->timing_launches = 0;
  return QK_OK;
}

int qk_accum_timing_read(qk_accum *a, double *total_ms, uint64_t *launches) {
  return qk_accum_timing_read_batch(a, total_ms, nullptr, launches);
}

int qk_accum_timing_read_range(qk_accum *a, double *hist_min_ms, double *hist_max_ms) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  int rc = drain_timing(a);
  if (rc) return rc;
  if (hist_min_ms) *hist_min_ms = a->timing_min_ms;
  if (hist_max_ms) *hist_max_ms = a->timing_max_ms;
  return QK_OK;
}

int qk_accum_timing_read_batch(qk_accum *a, double *hist_ms, double *batch_ms, uint64_t *launches) {
  if (!a) return fail(QK_EINVAL, "acc is NULL");
  int rc = drain_timing(a);
  if (rc) return rc;
  if (hist_ms) *hist_ms = a->timing_ms;
  if (batch_ms) *batch_ms = a->timing_batch_ms;
  if (launches) *launches = a->timing_launches;
  return QK_OK;
}

}  // extern "C"
