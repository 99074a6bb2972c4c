// qk_kernels.hip.h — hand-written gfx950 kernels for quack's per-read
// accumulation loop (reference: read_fastq, quack.c:193-221).
//
// Mapping ("chunk owner"): a read segment of a position tile is cut into
// 8-byte chunks; lane (ri, ch) of a workgroup owns chunk `ch` of the ri-th read
// of every iteration, i.e. ALWAYS the same 8 positions.  Consequences:
//   * global loads are one 4-byte-ALIGNED dwordx3 per array per lane: the
//     12-byte window that contains the lane's 8 bytes, realigned with two
//     v_alignbyte_b32.  Windows of neighbouring lanes overlap (L1 absorbs it;
//     HBM traffic is unchanged).  Measured on the bare load loop: unaligned
//     dwordx2 0.60 ms, aligned dwordx3 windows 0.50 ms, ideal aligned stream
//     0.47 ms (3 GB) — 150-byte reads are misaligned 3 times out of 4.
//     Addresses are 32-bit offsets from the workgroup's (scalar) slice base.
//   * quality scores go to an LDS histogram laid out [byte value][j&3][ch] with
//     two u16 counters per dword (j>>2 selects the half) and a row stride that
//     is a multiple of 32 dwords, so the LDS bank of an update is fixed by the
//     lane's chunk alone — measured SQ_LDS_BANK_CONFLICT ~0.  Per base:
//     v_bfe (byte -> row) + v_mad_u32_u24 (row -> address) + ds_add_u32;
//   * base content never touches the LDS in the loop: the owner of a position
//     keeps SWAR byte counters (not-T / not-C / not-G, 4 positions per VGPR) and
//     spills "events - count" every <= 255 steps.  A = valid - T - C - G at
//     flush time.  (Measured: LDS *instruction issue*, ~7 cycles per ds_add
//     wave-instruction per CU, is the binding resource, not the LDS array — two
//     atomics per base cost 2x one.)
//   * the step loop (issue = addresses + loads, consume = histogram) is
//     software-pipelined for fixed-length batches: the loads of step k+1 are in
//     flight while step k is consumed (template parameter PD).
//   * adapter first hit (ADAPT builds; quack.c:206-217): the SWAR indicators
//     give 2-bit codes, 8 of them packed per chunk; the 9 predecessor codes come
//     from lanes -1 / -2 by DPP wave_shr:1 (two feeder lanes per wave re-compute
//     the previous wave's last chunks; with several tiles two halo lanes per
//     read row cover the 16 positions before the tile).  Every 10-mer window
//     contains exactly one 9-mer that ends on an even position, so the lane
//     probes FOUR 9-mers per chunk (not eight windows) in a 2^18-bit filter at
//     LDS byte 0 that holds the prefix and the suffix 9-mer of every adapter
//     10-mer; what passes (~2 % of the lanes, almost all false positives) goes
//     into a per-wave LDS queue and is checked against the exact 2^20-bit table
//     a wave's worth at a time (checking on the spot would issue the check for
//     the whole wave in four steps out of five); atomicMin(first_hit[read]).
//   * long reads: position tiles x read slices; persistent workgroups pull
//     slices from per-tile device counters and flush only when they change
//     tile.  Batches whose reads start on 128-byte lines (AL) get tiles of whole
//     cache lines and aligned 8-byte loads.
// Quality rows are raw byte values (& 127); the mapping to quack's 91 score
// bins is applied once, at flush time (histograms are linear, so re-binning
// afterwards is exact).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qk {

constexpr int kQRows = 128;            // quality byte & 127
constexpr int kOutRows = 97;           // quack.c:134-139
constexpr int kRowContent = 91;
constexpr int kRowLength = 95;
constexpr int kRowKmer = 96;
constexpr uint32_t kMaxReadsPerSlice = 65535; // u16 counters: <=1 hit/read/pos
// bits of the device status word (HistParams::status), read back by qk_accum_sync
constexpr uint32_t kStatusNotAligned = 1u;   // a batch submitted as QK_BATCH_ALIGNED128 holds a read off a 128-byte line
constexpr uint32_t kStatusBadLength = 2u;    // a device-side lengths[] entry exceeds the stride / the table
constexpr uint32_t kStatusBadPads = 4u;      // a batch submitted as QK_BATCH_NEUTRAL_PADS holds a pad byte that is not 0xFF (checked on request)

// Device-side parameter block of one launch.
struct HistParams {
  const uint8_t *seq;
  const uint8_t *qual;
  const uint64_t *offsets;      // NULL => fixed-length batch; else start of every read (+ the end of the last
                                //   one when `lengths` is NULL: packed batch, read r ends where r+1 starts)
  const uint32_t *lengths;      // gapped batch: length of every read (NULL: packed)
  // Long ragged reads (several tiles): reads ordered by the number of tiles they
  // reach, descending, so that the reads of tile t are order[0 .. reach[t]) and a
  // far tile never looks at the reads that end before it (reach_* kernels below).
  // NULL: natural order, every tile scans every read.
  const uint32_t *order;
  const uint32_t *reach;
  uint32_t stage_reads;         // ragged: reads staged per pass (a multiple of 1024: short reads want long passes)
  uint32_t lengths_done;        // length_count / the kmers==NULL count were taken by ragged_length_kernel
  uint32_t *status;             // kStatus* bits: what a kernel found wrong with the batch (fails the next sync)
  uint32_t check_aligned;       // the batch was submitted as QK_BATCH_ALIGNED128 (whichever variant runs it)
  uint32_t len_limit;           // strided batches: the longest read the caller declared (lengths[] on the device are
                                //   checked against it by the length kernels; 0: nothing to check)
  unsigned long long *table;    // planar [kOutRows][table_len]
  uint32_t *table32;            // the same layout in 32-bit words, or NULL: what hist_kernel's flush adds to (quality and content
                                //   rows) — the L2's atomic units take ~1.5 TB/s of ADDED BYTES chip-wide (tools/atomic_rate.hip:
                                //   40 rows x 512 positions per workgroup, 28.5 us as u64, 12.9 us as u32), and a long-read workgroup
                                //   spends 6 % of its time there.  The host folds it into `table` before anyone reads that, and before
                                //   4 G reads could have gone into it.
  uint32_t *first_hit;          // per-read first adapter hit (ADAPT only)
  const uint32_t *kmer_bits;    // 2^20-bit exact table (ADAPT only)
  const uint32_t *kmer_filter;  // [2^18 bits: suffix 9-mers (separate scan kernel) | 2^18 bits: the fused path's filter, copied into LDS]
  uint64_t n_reads;
  uint64_t total_bytes;         // offsets[n_reads]; loads are clamped to it
  uint64_t reads_per_slice;     // <= kMaxReadsPerSlice
  uint32_t read_len;            // fixed-length batches
  uint32_t stride;              // fixed-length and strided batches: read r starts at r * stride.  == read_len for packed reads; a
                                //   multiple of 4 just above it for PADDED ones (round 4: uniform reads whose length is not a
                                //   multiple of 4 — 150, 250, 50 — laid out so that every chunk starts on a dword: the AL / W16
                                //   kernels then run them; the pad bytes count into columns >= read_len, which are never flushed)
  // Grouped rows (round 4; fixed-length reads with the fused adapter scan, 16 positions per lane, one tile): a ROW of the
  // batch is `group` consecutive reads, `gstride` bytes apart, and the kernel treats it as one long read of
  // (group - 1) * gstride + read_len positions — `stride` is then the distance between rows, n_reads counts rows.  A read of
  // 150 bases takes 10 lanes of 16 positions (6 % of them idle), two reads 152 bytes apart take 19; 100 bp: 7 lanes for
  // one read, 19 for three; 36 bp: 3 for one, 9 for four.  Only the rare paths know about it: the flush folds the column
  // groups onto positions, the spill and the candidate check ask which read a column belongs to.
  uint32_t fh_words;            // words of the first-hit ring in LDS (a power of two >= kFhRing; fused adapters, fixed length)
  uint32_t group;               // reads per row (>= 1)
  uint32_t gstride;             // bytes between the reads of a row (group > 1)
  uint32_t table_len;           // positions in `table`
  uint32_t n_tiles;             // position tiles
  uint32_t tile_pos;            // positions per tile (multiple of 8)
  uint32_t ch;                  // chunks per tile = tile_pos / 8
  uint32_t row_dwords;          // LDS row stride: 4*replicas*ch rounded up to 32 banks
  uint32_t replicas;            // column replicas (short reads: reads sharing a lane group get different banks)
  uint32_t halo;                // ADAPT with several tiles: 2 extra lanes per read row cover the 16 positions before the tile
  uint32_t reads_per_iter;      // chunk lanes per workgroup / ch
  uint32_t n_slices;            // read slices per tile; work items = n_tiles * n_slices
  uint32_t *queue;              // [n_tiles] slice counters (several tiles), or NULL
  // several tiles and the work of every tile known (reads sorted by reach): no queue — workgroup b
  // takes the b-th of gridDim.x equal, contiguous shares of the read-tiles
  const unsigned long long *tile_prefix;   // [n_tiles + 1] reads of the tiles before t, or NULL (t * n_reads)
  uint32_t static_split;
  uint32_t tile_overhead;       // static split: what a tile costs a workgroup besides its reads (the flush), in read-tiles
  uint32_t no_adapters;         // kmers == NULL semantics (quack.c:210,215)
  uint32_t count_in_kernel;     // fused adapters, one tile: hist_kernel resets first_hit[] and takes the kmer_count itself
  // exact LDS-resident membership table the queued candidates are checked against (0: the
  // global 2^20-bit table — huge adapter files, or no room): 2^bucket_log2 buckets of eight
  // u16 remainders
  const uint4 *kmer_buckets;
  uint32_t bucket_log2;
  uint32_t bucket_mul;          // odd multiplier: km' = km * mul mod 2^20 is a bijection
};

// word >> (byte B of w, low five bits): the shift amount comes out of the register's byte by sub-dword addressing
template <int B>
__device__ __forceinline__ uint32_t shr_by_byte(uint32_t word, uint32_t w) {
  uint32_t r;
  if constexpr (B == 0)
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(word));
  else
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(word));
  return r;
}

// 12 bytes from a 4-byte-aligned address: one global_load_dwordx3
struct u32x3 { uint32_t x, y, z; };
// 8 bytes from an 8-byte-aligned address (batches whose reads start on cache
// lines: the chunk is aligned as it is)
struct u32x2 { uint32_t x, y; };   // (4-byte alignment: one global_load_dwordx2 all the same)
__device__ __forceinline__ u32x3 load8_aligned(const uint8_t *p) {
  const u32x2 v = *reinterpret_cast<const u32x2 *>(__builtin_assume_aligned(p, 4));
  return u32x3{v.x, v.y, 0u};
}
__device__ __forceinline__ u32x3 load12_aligned(const uint8_t *p) {
  return *reinterpret_cast<const u32x3 *>(__builtin_assume_aligned(p, 4));
}

// 16 bytes from a 4-byte-aligned address: one global_load_dwordx4 (W16 builds: a lane owns two adjacent chunks)
struct u32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u32x4 load16_aligned(const uint8_t *p) {
  return *reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(p, 4));
}

template <bool W16> struct load_reg { using type = u32x3; };
template <> struct load_reg<true> { using type = u32x4; };

// the 8 bytes that start `s` (0..3) bytes into a 12-byte window
__device__ __forceinline__ uint2 window8(u32x3 w, uint32_t s) {
  return make_uint2(__builtin_amdgcn_alignbyte(w.y, w.x, s), __builtin_amdgcn_alignbyte(w.z, w.y, s));
}

__device__ __forceinline__ void lds_add(uint32_t *lds, uint32_t byte_off,
                                        uint32_t val) {
  uint32_t *p = reinterpret_cast<uint32_t *>(
      reinterpret_cast<char *>(lds) + byte_off);
  __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Per-byte "!= K" indicator (0/1 in each byte) on the 5-bit letter key.
// quack.c:148-150,201 maps letters by (c-65)&~32: T->1, C->2, G->3, every
// other letter of its defined domain (A..T, a..t) -> 0.  (c & 31) is injective
// on that domain: T=20, C=3, G=7.  Bytes outside it index lookup[] out of
// bounds in the reference (undefined); here they alias the letter with the
// same low five bits.  The kernels count "not the letter" (one instruction
// shorter than the equality) and take events - count when they spill.
__device__ __forceinline__ uint32_t swar_ne(uint32_t w, uint32_t k4) {
  const uint32_t y = (w & 0x1F1F1F1Fu) ^ k4;
  const uint32_t z = y + 0x7F7F7F7Fu;
  return (z >> 7) & 0x01010101u;
}

constexpr uint32_t kKeyT = 0x14141414u, kKeyC = 0x03030303u, kKeyG = 0x07070707u;

// LDS image (dwords): quality histogram [128][row_dwords] | base counters
// [4: valid,T,C,G][8*ch] | length_count [8*ch] | misc[4]
// Bank balance of the quality counters.  An LDS atomic of a wave costs the CU
// 2 cycles x (the most lanes that meet in one of the 32 banks) — measured,
// tools/lds_rate.hip: 4.2 cycles at two lanes per bank, 7.3 at four, 128 when all
// 64 hit one address — and a wave holds the chunk lanes of several reads, which
// all want the same columns.  So the columns exist as S = 4 R "sets" of CH
// dwords, laid side by side (set s at dword s * CH): the lanes of read row ri
// use set (ri + i) % S in their i-th counting instruction, which makes the 64
// lanes of a wave touch 64 CONSECUTIVE dwords of the row (two lanes per bank; a
// third where the window wraps around the end of the sets).  Set s counts byte
// s % 4 of a quality dword (positions s % 4 and s % 4 + 4 of the chunk in the
// two u16 halves), so a lane takes its bytes in rotated order — one v_alignbit
// per dword — and only R replicas of the counters exist, not S; the flush sums
// the replicas.  R = 40 / CH keeps a row within 160 dwords (80 KiB), which is
// what fits next to the adapter tables; wider tiles have R = 1.
// (lanes per bank, worst over the wave, mean over waves and instructions:
//  150 bp 3.9 -> 2.4, 300 bp 3.4 -> 2.3, 100 bp 5.0 -> 2.4, 36 bp 3.2 -> 2.0;
//  the ideal is 2.0)
inline __host__ __device__ uint32_t hist_replicas(uint32_t ch) { return ch >= 40u ? 1u : 40u / (ch ? ch : 1u); }
inline __host__ __device__ uint32_t hist_row_dwords(uint32_t ch, uint32_t replicas) {
  return (4u * replicas * ch + 31u) / 32u * 32u;
}
constexpr uint32_t kFusedFilterLog2 = 18;   // 2^18-bit 9-mer filter = 32 KiB of LDS
constexpr uint32_t kFusedFilterWords = (1u << kFusedFilterLog2) / 32u;
constexpr uint32_t kNoHit = 0xFFFFFFFFu;
constexpr uint32_t kStageReads = 1024;      // ragged: read descriptors staged in LDS per pass (at least; HistParams::stage_reads)
constexpr uint32_t kStageReadsMax = 8192;
// Candidate queue of the fused adapter path: what passes the 9-mer filter (a lane
// or two per wave and step, nearly all false positives) is not checked on the
// spot — that would issue the check's instructions for the whole wave, and the
// kernel is VALU-bound — but appended to a per-wave queue in LDS and checked
// against the exact table when the queue holds a wave's worth of entries.
constexpr uint32_t kCandCap = 96;                       // entries per wave: drained above 32, a step adds <= 64
constexpr uint32_t kCandWords = 16u * kCandCap * 2u;    // 16 waves x 96 entries x 8 bytes = 12 KiB
// W16 builds: one 16-byte entry per LANE (16 positions).  Round 4: 64 entries per wave, checked when a step's entries would
// not fit any more — nearly a full wave's worth per check (the rule above ran the checks of 25 % spliced 150 bp reads with 33-45 of
// their 64 lanes) — and 16 KiB instead of 24: the 8 KiB go to the first-hit ring (HistParams::fh_words)
constexpr uint32_t kCandCap16 = 64;
constexpr uint32_t kCandWords16 = 16u * kCandCap16 * 4u;
// First hits of a fixed-length one-tile batch stay in the LDS (round 3): all lanes of a read sit in one workgroup in
// one step, so min(first hit) needs no global memory — a ring of fh_words (>= kFhRing) words indexed by the read's place in the
// slice, folded into the kmer_count row and cleared every G <= fh_words / 2 reads behind a workgroup barrier (every
// wave drains its queue in front of the barrier).  What it replaces: one global atomicMin per read with a hit — 2.9 M
// per 10M x 300, each a random 128-byte line of a 40 MB array through the L2 and, gfx9 having ONE counter for loads
// and atomics, inside the wait for the next step's loads — 7-10 % of the kernel, measured by leaving the atomic out;
// plus the reset of that array before and its read-back after the loop.
constexpr uint32_t kFhRing = 2048;      // the smallest ring (what fits beside a 304-position histogram); HistParams::fh_words is the launch's
constexpr uint32_t kFhRingMax = 16384;  // (round 4: the planner takes the largest power of two the LDS has room for — a fold every
                                        //  fh_words / 2 reads costs every wave a drain of its queue, however empty, and a barrier)
// WIDE layout (round 5) of the one VALU-bound variant — fixed-length reads, fused adapter scan, 16 positions per lane.  A lane
// owns a PAIR of chunks, and the u16-pair counters of a pair's byte-b columns (set s) are the two dwords `pair` and `pair + 32`
// of a 64-dword row; the quality counter of (row q, pair index p, odd chunk o) lives at LDS byte
//     (p / 32) * 65536 + q * 256 + (p % 32) * 4 + o * 128
// so the row is BYTE 1 of the address: an address register is a per-lane constant (plane in byte 2, column in byte 0) that serves
// both chunks of the lane (the odd one through the instruction's offset field), and one SDWA instruction per base —
//   v_and_b32_sdwa addr, 0x7f, w  dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src1_sel:BYTE_j
// — extracts the quality byte, masks it and puts it in place (the narrow layout: shift + v_bitop3, two per base and eight address
// registers; here 36 -> 20 VALU per 16 positions with the rotation, and four registers).  Bank = column as before.  A plane (64
// columns x 128 rows) is 32 KiB and planes lie 64 KiB apart — bit 15 of an address is part of byte 1 and must stay clear —, which
// leaves two 32 KiB gaps: the 9-mer filter sits in the first (a probe's address is key field + 32768, the constant goes into the
// instruction's offset field), the candidate queues, the letter / length / kmer rows and the bucket table in the second.  The
// first-hit ring takes the pairs of the last plane that no counter uses (128 rows x 32 or 16 words) when it fits there.
constexpr uint32_t kWidePlaneStride = 16384u;       // dwords between planes (64 KiB)
constexpr uint32_t kWideFilterOff = 8192u;          // dwords: [32 KiB, 64 KiB)
constexpr uint32_t kWideRestOff = 24576u;           // dwords: [96 KiB, 128 KiB) queue | letter rows | misc | buckets
constexpr uint32_t kWideTopOff = 32768u;            // dwords: from 128 KiB on (fewer than three planes): buckets | ring
__host__ __device__ constexpr uint32_t qhist_index_wide(uint32_t row, uint32_t pair, uint32_t odd) {
  return (pair >> 5) * kWidePlaneStride + row * 64u + (pair & 31u) + odd * 32u;
}
inline __host__ __device__ uint32_t wide_planes(uint32_t cols) { return (cols + 63u) / 64u; }   // cols = 4 R CH = two per pair
// words of a first-hit ring that fit into the unused pairs of the last plane: W = 32 or 16 words per row (the last W / 2
// pairs): word i lives in row i / W, at dword 32 - W / 2 + (i % W) % (W / 2) + 32 * ((i % W) / (W / 2))
inline __host__ __device__ uint32_t wide_spare_ring(uint32_t cols) {
  const uint32_t free_pairs = wide_planes(cols) * 32u - cols / 2u;
  return free_pairs >= 16u ? 4096u : free_pairs >= 8u ? 2048u : 0u;
}
// Where the bucket table and the first-hit ring of a wide launch go (host and kernel compute the same): the bucket table behind
// the small rows in the second gap if it fits there, else behind the last plane's slot; the ring in the last plane's unused pairs
// if it fits there, else behind the last plane's slot (fewer than three planes), else in the second gap.
struct WidePlan {
  uint32_t planes;
  uint32_t bucket_off;   // dwords from LDS byte 0
  uint32_t ring_off;     // dwords from LDS byte 0; 0: in the unused pairs of the last plane
  uint32_t bytes;        // of dynamic LDS to ask for; 0: the shape does not fit
};
inline __host__ __device__ WidePlan wide_plan(uint32_t ch, uint32_t replicas, uint32_t bucket_log2, uint32_t fh_words) {
  WidePlan w{};
  const uint32_t cols = 4u * replicas * ch;
  w.planes = wide_planes(cols);
  if (w.planes > 3u) return w;
  uint32_t gap_at = kWideRestOff + kCandWords16 + 6u * 8u * ch + 4u;   // queue | [4][TP] letters | length | kmer | misc
  const uint32_t gap_end = kWideTopOff;
  uint32_t top_at = kWideTopOff;
  const uint32_t top_end = w.planes == 3u ? kWideTopOff : kWideTopOff + 8192u;   // (three planes: the last one lives there)
  if (gap_at > gap_end) return w;
  const uint32_t bw = bucket_log2 ? (4u << bucket_log2) : 0u;
  if (bw <= gap_end - gap_at) {
    w.bucket_off = gap_at;
    gap_at += bw;
  } else if (bw <= top_end - top_at) {
    w.bucket_off = top_at;
    top_at += bw;
  } else {
    return w;
  }
  if (fh_words <= wide_spare_ring(cols)) {
    w.ring_off = 0;
  } else if (fh_words <= top_end - top_at) {
    w.ring_off = top_at;
    top_at += fh_words;
  } else if (fh_words <= gap_end - gap_at) {
    w.ring_off = gap_at;
    gap_at += fh_words;
  } else {
    return w;
  }
  w.bytes = (w.planes == 3u ? kWideTopOff + 8192u : top_at) * 4u;
  return w;
}
inline size_t hist_lds_bytes(uint32_t ch, uint32_t replicas, bool adapt = false, uint32_t bucket_log2 = 0, bool ragged = false,
                             uint32_t stage_reads = kStageReads, bool w16 = false, uint32_t fh_words = kFhRing) {
  if (adapt && w16 && !ragged) {   // the wide variant (hist_body: WIDE)
    const size_t b = wide_plan(ch, replicas, bucket_log2, fh_words).bytes;
    return b ? b : (size_t)1 << 30;   // (does not fit: larger than any LDS)
  }
  return ((size_t)kQRows * hist_row_dwords(ch, replicas) + (adapt ? 6u : 5u) * 8u * ch + 4u + (adapt ? kFusedFilterWords + (w16 ? kCandWords16 : kCandWords) : 0u)) * sizeof(uint32_t) +
         (adapt && bucket_log2 ? ((size_t)16 << bucket_log2) : 0) + (ragged ? (size_t)stage_reads * (adapt ? 12 : 8) : 0) +
         (adapt && !ragged ? (size_t)fh_words * 4 : 0);   // fixed-length batches: the first-hit ring (where a ragged batch stages its reads)
}

// The fused path works on COMPLEMENTED 2-bit codes (3 - code: the "not T / not C /
// not G" indicators give them without an inversion), so its keys are the
// reference's 10-mer indices (quack.c:150,208) xor 0xFFFFF.
constexpr uint32_t kKmerMask = 0xFFFFFu;
#ifdef QK_TIMING   /* experiment: where a workgroup's time goes (100 MHz stamps of thread 0; qk_shim prints them at finish) */
__device__ unsigned long long qk_timing[1024 * 16];
__device__ unsigned long long qk_wtime[1024 * 16];   /* per wave: when its step loop ended */
#define QK_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024u) qk_timing[blockIdx.x * 16u + (i)] = wall_clock64(); } while (0)
#else
#define QK_MARK(i) do { } while (0)
#endif
// Exact membership in the bucket table: km' = km*mul mod 2^20 (a bijection for
// odd mul); the top bucket_log2 bits pick a 16-byte bucket, the rest (< 2^15)
// is stored with bit 15 set.  Empty slots are 0.
__device__ __forceinline__ bool bucket_has(const uint4 *buckets, uint32_t km, uint32_t mul, uint32_t log2b) {
  const uint32_t h = (km * mul) & 0xFFFFFu;
  const uint32_t rem = (h & ((1u << (20u - log2b)) - 1u)) | 0x8000u;
  const uint4 b = buckets[h >> (20u - log2b)];
  const uint32_t r2 = rem | (rem << 16);
  const uint32_t x0 = b.x ^ r2, x1 = b.y ^ r2, x2 = b.z ^ r2, x3 = b.w ^ r2;
  // a 16-bit half of x is zero <=> match
  auto zero_half = [](uint32_t x) { return ((x & 0xFFFFu) == 0u) | ((x >> 16) == 0u); };
  return zero_half(x0) | zero_half(x1) | zero_half(x2) | zero_half(x3);
}

// Byte at an absolute LDS address.  The adapter kernels keep their window
// filter at LDS byte 0 (first thing in the dynamic segment; they have no static
// LDS, which launch_hist checks), so a probe's address is the key field itself
// — the address of an `extern __shared__` array is a link-time constant the
// compiler would otherwise add per probe.
typedef const __attribute__((address_space(3))) uint8_t lds_const_u8;
__device__ __forceinline__ uint32_t lds_abs_u8(uint32_t byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(lds_const_u8 *)byte_addr;
#else
  return byte_addr & 0u;   // host pass of the single-source compile: never called
#endif
}
// ... and the dword at an absolute, 4-byte-aligned LDS address.  The fused scan probes its filter this way (round 3):
// bit (key & 31) of the dword at (key >> 5) * 4 IS bit (key & 7) of the byte at key >> 3, so the layout is the same,
// but the probe is  v_lshrrev + v_and (address), ds_read_b32, v_bfe_i32 word, key, 1  — the bit index is the low
// five bits of the key as it stands, where the byte probe needed a v_bfe for the address, a v_bfe for the bit index
// and (the compiler's) v_and 0xff on what ds_read_u8 returned.
typedef const __attribute__((address_space(3))) uint32_t lds_const_u32;
__device__ __forceinline__ uint32_t lds_abs_u32(uint32_t byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(lds_const_u32 *)byte_addr;
#else
  return byte_addr & 0u;
#endif
}

// ---- quality histogram layout and address -----------------------------------
// The u16-pair counters of a tile are kept as PLANES of 128 rows x 32 dwords:
// column c (0 .. row_dwords) of quality row b lives at dword
//     (c / 32) * 4096 + b * 32 + (c % 32)
// so that bank == column (as with one long row per quality value) AND the byte
// address is  (b << 7) | lane_constant  with disjoint bit fields.  On gfx950 the
// VALU has two cost classes (tools/instr_rate.hip): v_lshrrev by a constant,
// v_and/v_or and v_bitop3 on VGPRs issue at full rate, while v_bfe, v_mad_u32_u24,
// every left shift and every VOP3 with an SGPR/literal operand take ~1.7x as
// long.  The address of byte J of a quality dword is therefore
//     J >= 1:  v_lshrrev(8J-7) ; v_bitop3 (x & 0x3F80) | qcol      (2 fast ops)
//     J == 0:  v_and 0x7F ; v_lshl_or 7                            (1 fast, 1 slow)
// instead of v_bfe + v_mad_u32_u24 (2 slow ops) per base.
constexpr uint32_t kPlaneDwords = 128u * 32u;
__host__ __device__ constexpr uint32_t qhist_index(uint32_t row, uint32_t col) {
  return (col >> 5) * kPlaneDwords + row * 32u + (col & 31u);
}
typedef __attribute__((address_space(3))) uint32_t lds_u32;
template <int J, uint32_t BASE>
__device__ __forceinline__ void qhist_add(uint32_t w, uint32_t mask7, uint32_t qcol, uint32_t val) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t addr;
#ifdef QK_QADDR_SLOW   // A/B: the two-slow-op address (v_bfe + v_mad_u32_u24) on the same layout
  addr = __umul24(__builtin_amdgcn_ubfe(w, 8 * J, 7), 128u) + qcol;
  (void)mask7;
#else
  if (J == 0) {
    const uint32_t t = w & 0x7Fu;
    asm("v_lshl_or_b32 %0, %1, 7, %2" : "=v"(addr) : "v"(t), "v"(qcol));
  } else {
    const uint32_t s = w >> (8 * J - 7);
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(addr) : "v"(s), "v"(mask7), "v"(qcol));
  }
#endif
  // absolute LDS address: BASE (bytes, a compile-time constant) goes into the
  // instruction's offset field
  __hip_atomic_fetch_add((lds_u32 *)addr + BASE / 4u, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  (void)w; (void)mask7; (void)qcol; (void)val;
#endif
}

// wide layout (qhist_index_wide): the quality row is byte 1 of the counter's address, so ONE instruction takes byte J of the
// quality dword, masks it to 7 bits and puts it into the lane's address register, whose other bytes (plane, column) stay
template <int J, uint32_t OFF>
__device__ __forceinline__ void qhist_add_wide(uint32_t w, uint32_t m7f, uint32_t &addr, uint32_t val) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(J >= 0 && J < 4, "byte of a dword");
#ifdef QK_WIDE_NOSDWA   /* experiment: the same layout, the address by plain instructions */
  {
    const uint32_t a2 = (((w >> (8 * J)) & 0x7Fu) << 8) | (addr & 0xFFFF00FFu);
    __hip_atomic_fetch_add((lds_u32 *)a2 + OFF / 4u, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    (void)m7f;
    return;
  }
#endif
  if constexpr (J == 0) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0" : "+v"(addr) : "v"(m7f), "v"(w));
  if constexpr (J == 1) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_1" : "+v"(addr) : "v"(m7f), "v"(w));
  if constexpr (J == 2) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2" : "+v"(addr) : "v"(m7f), "v"(w));
  if constexpr (J == 3) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3" : "+v"(addr) : "v"(m7f), "v"(w));
  __hip_atomic_fetch_add((lds_u32 *)addr + OFF / 4u, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  (void)w; (void)m7f; (void)addr; (void)val;
#endif
}

// value of the same register in lane-1 (v_mov_b32_dpp wave_shr:1); lane 0 gets 0
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);
}

// MODE: 0 full; 1 loads only; 2 quality only; 3 bases only (ablation builds
// used by tools/kbench only; the shim always launches MODE 0).
//
// Work items = (tile, read slice).  One tile: block b owns slice b.  Several
// tiles (p.queue set): persistent workgroups pull slices from per-tile device
// counters and only flush / clear their LDS histogram when they change tile (or
// their u16 counters could overflow), so the number of table flushes is
// ~ tiles + workgroups instead of tiles x slices.
#ifndef QK_MIN_WAVES_PER_SIMD
#define QK_MIN_WAVES_PER_SIMD 1   // experiments: (T/256)*k asks for k workgroups per CU
#endif
// AL, ragged: every read of the batch starts on a 128-byte boundary, so a
// lane's chunk is 8-byte aligned and a 512-position tile of a read is exactly
// four cache lines (see QK_BATCH_ALIGNED128 in quack_hip.h).
// AL, fixed length: read_len is a multiple of 4 (36, 76, 100, 300 ...), so every
// chunk starts on a dword: one global_load_dwordx2, no 12-byte window, no
// v_alignbyte.
// SV (fixed stride, variable length; FIXED and AL too): read r occupies
// [r * stride, r * stride + lengths[r]) with stride = p.read_len a multiple of 4 —
// the form the host feed gives trimmed short reads (most of them full length): the
// addresses, the pipelined step loop and the dword loads of a fixed-length batch,
// plus the tail masks of a ragged one.  The length of a read travels with its
// loads (one more dword in flight per read); length_count comes from
// ragged_length_kernel.
// (SV builds are held to 64 VGPRs: with the tail masks the loop is VALU-heavier than
// the fixed-length one and gains from two workgroups per CU, 0.61 -> 0.53 ms per 10M
// trimmed 150 bp reads; the plain fixed-length kernel is faster with one)
// W16 (round 3; AL builds only): a lane owns TWO adjacent chunks — 16 positions — of a read and fetches them with one
// global_load_dwordx4 per array.  Measured on loads alone (tools/piece_rate.hip): the long-read pattern — a wave takes
// a read's 512-byte piece of the tile, the next read lies ~10 kb away — moves 4.8 TB/s with 8 bytes per lane and 5.25
// with 16 (a contiguous stream: 5.9), whatever the piece size; and every per-lane cost of a step (addresses, the DPP
// for the predecessor codes, the candidate vote) is paid once per 16 positions.  The LDS image keeps its 8-position
// chunk layout; only the column order changes (even chunks first, so that the lanes of one counting instruction
// still touch consecutive dwords).
// (the long-read build — ragged, W16, no adapter scan — is held to 120 VGPRs: four of its waves then leave 32 registers of
// a SIMD free, which is what the waves of the NEXT batch's reach pre-pass (12-16 VGPRs, 2 KiB of LDS) need to run beside it
// on the side stream instead of waiting for a workgroup to retire)
__device__ __forceinline__ void wave_count_lds(uint32_t *cnt, uint32_t key, bool valid);   // (below: equal keys of a wave added up first)

// ONE (round 5): the launch has one position tile and no work queue — block b owns read slice b, nothing else is compiled in.
// The VALU-bound variant (fixed length + adapter scan + 16 positions per lane) runs at the register file's limit (128 VGPRs for
// four waves per SIMD); with the three work-loop forms inlined side by side the one-tile form's step loop spilled its counter
// addresses to scratch (s_waitcnt vmcnt(0) in front of every LDS atomic: the kernel ran at half speed).
template <int T, int U, bool FIXED, int MODE, bool ADAPT = false, int PD = 1, bool AL = false, bool SV = false, bool W16 = false, bool NP = false, bool ONE = false>
__device__ __forceinline__ void hist_body(const HistParams &p) {
  static_assert(!NP || SV, "neutral pads are a property of strided batches");
  static_assert(!SV || (FIXED && AL), "strided batches are a variant of the dword-aligned fixed-length path");
  static_assert(!W16 || (AL && MODE == 0 && (!SV || (NP && ADAPT && FIXED))),
                "16 positions per lane: dword-aligned batches; strided ones with neutral pads and the adapter scan only");
  constexpr bool STAGED = !FIXED;   // ragged batches: the read list is staged in LDS pass by pass
  constexpr int K = W16 ? 2 : 1;    // 8-position chunks per lane
  using LoadT = typename load_reg<W16>::type;
  constexpr uint32_t kCandWordsT = W16 ? kCandWords16 : kCandWords;
  extern __shared__ uint32_t lds_raw[];
  // the wide LDS layout of the VALU-bound variant (see qhist_index_wide): counter planes 64 KiB apart, the filter and the
  // small rows in the gaps
  constexpr bool WIDE = W16 && ADAPT && FIXED && MODE == 0;
  // ADAPT: the window filter sits first, so that the probes' LDS addresses are
  // "field + constant" (no per-probe add of a layout-dependent base)
  uint32_t *lds = lds_raw + ((ADAPT && !WIDE) ? kFusedFilterWords + kCandWordsT : 0u);
  const uint32_t tid = threadIdx.x;
  const uint32_t CH = p.ch;
  const uint32_t RD = p.row_dwords;
  const uint32_t TP = 8u * CH;  // == p.tile_pos
  const uint32_t wide_cols = 4u * p.replicas * CH, wide_np = wide_planes(wide_cols);
  const uint32_t hist_words = kQRows * RD;   // (narrow layout)
  uint32_t *lds_base = WIDE ? lds_raw + kWideRestOff + kCandWords16 : lds + hist_words;  // [4][TP]
  uint32_t *lds_len = lds_base + 4u * TP;
  uint32_t *lds_kmer = lds_len + TP;      // ADAPT: kmer_count of the tile (count_in_kernel)
  uint32_t *lds_misc = lds_kmer + (ADAPT ? TP : 0u);      // [0] reads longer than 10, [1] next item
  uint32_t *lds_filter = WIDE ? lds_raw + kWideFilterOff : lds_raw;         // ADAPT only
  const uint8_t *filt8 = reinterpret_cast<const uint8_t *>(lds_raw);   // == LDS byte 0, see lds_abs_u8
  (void)filt8;
  // ragged batches: descriptors {start - slice base, length [| index << 16]} of
  // the reads of the current pass that reach this tile, compacted
  // (the filter, hist words, 5*TP and 4 are all multiples of 4 dwords: 16-byte aligned)
  // (wide: wide_plan says where the bucket table and the ring go)
  const WidePlan wplan = WIDE ? wide_plan(CH, p.replicas, p.bucket_log2, p.fh_words) : WidePlan{};
  uint4 *lds_buckets = reinterpret_cast<uint4 *>(WIDE ? lds_raw + wplan.bucket_off : lds_misc + 4u);
  uint2 *lds_list = reinterpret_cast<uint2 *>(
      WIDE ? reinterpret_cast<char *>(lds_raw + wplan.ring_off)
           : reinterpret_cast<char *>(lds_misc + 4u) + ((ADAPT && p.bucket_log2) ? (16u << p.bucket_log2) : 0u));
  const uint32_t SR = STAGED ? p.stage_reads : kStageReads;
  uint32_t *lds_ridx = reinterpret_cast<uint32_t *>(lds_list + SR);   // ADAPT: index of a staged read within its pass
  uint32_t *lds_fh = reinterpret_cast<uint32_t *>(lds_list);          // ADAPT, FIXED: the first-hit ring (HistParams::fh_words words)

  const uint64_t TL = p.table_len;
  // grouped rows (HistParams::group): built into the one variant the planner uses them with
  constexpr bool GROUPS = W16 && ADAPT && FIXED;
  const uint32_t GRP = GROUPS ? p.group : 1u;
  const uint32_t GS = GROUPS ? p.gstride : 0u;
  const uint32_t row_len = FIXED ? (GRP - 1u) * GS + p.read_len : 0u;   // positions of a row that hold reads (SV: the stride)
  // column c of a row -> the read of the row it belongs to and the position in that read
  auto col_split = [&](uint32_t c, uint32_t &g, uint32_t &pos) {
    g = 0u;
    pos = c;
    if constexpr (GROUPS) {
      if (GRP > 1u) {
        g = c / GS;
        pos = c - g * GS;
      }
    }
  };

  QK_MARK(0);
#if defined(QK_ABL) && (QK_ABL & 1024)   /* experiment: what a launch costs without loading the adapter tables */
  if (false) {
#else
  if (ADAPT) {
#endif
    for (uint32_t i = tid; i < kFusedFilterWords; i += T) lds_filter[i] = p.kmer_filter[kFusedFilterWords + i];
    if (p.bucket_log2)
      for (uint32_t i = tid; i < (1u << p.bucket_log2); i += T) lds_buckets[i] = p.kmer_buckets[i];
  }

  // ADAPT: lanes 0 and 1 of every wave are feeders: they recompute the chunks
  // of the previous wave's lanes 62/63 so that lanes 2/3 find their
  // predecessors' codes by DPP; they take no part in the histograms.
  // (W16: one feeder lane, the previous wave's lane 63 — 16 positions either way)
  const uint32_t lane_id = tid & 63u;
  const uint32_t lane10 = lane_id << 10;
  const uint32_t feeders = ADAPT ? (W16 ? 1u : 2u) : 0u;
  const int32_t slot_signed = (int32_t)((tid >> 6) * (64u - feeders) + lane_id) - (int32_t)feeders;
  const uint32_t slot = slot_signed < 0 ? 0u : (uint32_t)slot_signed;
  // a read row is CH chunk lanes, preceded (ADAPT, several tiles) by two halo
  // lanes that cover the 16 positions in front of the tile: they load and
  // encode like any lane, so that the tile's first chunks find their
  // predecessor codes in lanes -1 / -2, but they count nothing
  const uint32_t H = ADAPT ? p.halo : 0u;            // (W16: one halo lane)
  const uint32_t CHW = CH / K + H;                   // lanes per read row
  const uint32_t ri = slot / CHW;
  const uint32_t chh = slot - ri * CHW;
  const bool is_halo = chh < H;
  const uint32_t chl = is_halo ? 0u : chh - H;       // the lane's place in its read row (halo lanes only ever add pads)
  const int32_t ch_signed = (int32_t)chh - (int32_t)H;
  const uint32_t RW = p.reads_per_iter;
  const bool lane_on = lane_id >= feeders && !is_halo && ri < RW;
  // one tile, fixed length: first hits are kept in lds_fh, see kFhRing (wave-uniform, the same for the whole launch;
  // a step of more reads than half the ring — reads of a chunk or two — keeps the global words)
  const uint32_t FHW = (ADAPT && FIXED) ? p.fh_words : kFhRing, FHM = FHW - 1u;
  const bool fh_ring = ADAPT && FIXED && p.count_in_kernel != 0 && RW * (uint32_t)U * GRP <= FHW / 2u;
  // wide layout: the ring lives in the columns of the last counter plane that no counter uses, fh_w (32 or 16) words per row
  const uint32_t fh_w = (WIDE && wplan.ring_off == 0u) ? (wide_spare_ring(wide_cols) >> 7) : 0u;
  const uint32_t fh_sh = fh_w == 32u ? 5u : 4u;
  auto fh_at = [&](uint32_t i) -> uint32_t * {   // word i (< FHW) of the ring
    if (WIDE && fh_w) {
      const uint32_t w = i & (fh_w - 1u), half = fh_w >> 1;   // the row's free pairs: dwords [32 - half, 32) and the same + 32
      return lds_raw + (wide_np - 1u) * kWidePlaneStride + ((i >> fh_sh) << 6) + (32u - half) + (w & (half - 1u)) + ((w >> (fh_sh - 1u)) << 5);
    }
    return lds_fh + i;
  };
  // rows between two folds of the ring: whole steps, their reads at most half the ring
  const uint32_t fh_group = fh_ring ? (FHW / 2u / GRP) / (RW * (uint32_t)U) * (RW * (uint32_t)U) : 0u;
  constexpr bool kScalarLoop = W16 && ADAPT && FIXED;   // see the step loop
  uint32_t fh_folded = 0, fh_next = 0xFFFFFFFFu;   // reads of the slice whose ring entries have been folded; the next fold point (wave-uniform)
  // the lane's K chunks of the tile and their LDS columns.  W16: chunks 2*chl and 2*chl + 1; the even chunks of a
  // tile take the first CH/2 columns of a set, the odd ones the second half — the lanes of ONE counting instruction
  // (same k) then touch consecutive dwords, as they do with one chunk per lane
  auto lds_col = [&](uint32_t c) { return W16 ? (c >> 1) + (c & 1u) * (CH / 2u) : c; };
  uint32_t chk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) chk[k] = (uint32_t)K * chl + (uint32_t)k;
  // byte address, in quality row 0, of the counter column this lane's i-th counting
  // instruction adds to (see hist_replicas): set (ri + i) % S, which holds byte
  // (ri + i) % 4 of the quality dword — byte i after a rotation by rot8 bits
  uint32_t qcol[K][4];
  const uint32_t R = p.replicas;
  const uint32_t kset = ri % (4u * R);
  const uint32_t rot8 = 8u * (kset & 3u);
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t set = (kset + i) % (4u * R);
      // (wide: one register per set serves both chunks — the odd one 128 bytes further on —, byte 1, the row, is rewritten for every base)
      qcol[k][i] = (WIDE ? qhist_index_wide(0u, set * (CH / 2u) + chl, 0u) : qhist_index(0u, set * CH + lds_col(chk[k]))) * 4u;
    }
  constexpr uint32_t kHistBase = (ADAPT && !WIDE) ? (kFusedFilterWords + kCandWordsT) * 4u : 0u;   // == (char*)lds - LDS byte 0
  uint32_t mask7;   // 127 << 7 in a VGPR (an SGPR or literal operand would put v_bitop3 in the slow class); wide: 127
  if constexpr (WIDE) asm("v_mov_b32 %0, 0x7f" : "=v"(mask7));
  else asm("v_mov_b32 %0, 0x3f80" : "=v"(mask7));
  const uint32_t one_lo = 1u, one_hi = 65536u;

  // ADAPT: this wave's candidate queue (see kCandCap) and its fill (wave-uniform)
  uint2 *cand_q = reinterpret_cast<uint2 *>(lds_raw + kFusedFilterWords) + (tid >> 6) * kCandCap;
  uint4 *cand_q16 = reinterpret_cast<uint4 *>(lds_raw + (WIDE ? kWideRestOff : kFusedFilterWords)) + (tid >> 6) * kCandCap16;   // W16: 16-byte entries
  uint32_t cand_n = 0;
  uint32_t n_gt10 = 0;        // reads longer than 10 (kmers==NULL path, quack.c:215)
  uint32_t fixed_reads = 0;   // FIXED, tile 0: reads seen since the last flush
  uint32_t keep = 0;          // MODE 1 only
  // SWAR byte counters for the 8 positions of a chunk: [0] positions 0-3, [1] 4-7 (W16: chunk k at [2k], [2k+1])
  uint32_t acc_v[2 * K] = {}, acc_t[2 * K] = {}, acc_c[2 * K] = {}, acc_g[2 * K] = {};
  uint32_t since_spill = 0;
  uint32_t prio_step = 0;     // steps this wave has consumed (whose turn it is at which issue priority, see consume)
  // acc_t/c/g count the bytes that are NOT T/C/G (swar_ne, one instruction
  // shorter than the equality) and acc_v the events in which a byte was masked;
  // the spill takes events - count.  Masked bytes are "not equal" in every
  // event, so they come out as zero.  `events` = accumulated steps since the
  // last spill (wave-uniform).  Fixed-length batches without the adapter scan
  // go one step further: the tail is a per-lane constant (fixed_mask), lanes past
  // the end of the slice sit out under the exec mask and count their own
  // events (steps_v), and no per-event valid counter is needed at all.
  // NP (round 4; strided batches whose pad bytes — behind a read's last base, up to the stride — are 0xFF, as the host feed
  // writes them): a pad byte counts into quality row 127, which the flush discards, and matches none of T / C / G, so the
  // strided kernel needs no tail masks either; what a position's `valid` count is — content[A] = valid - T - C - G — comes
  // from the lengths the step loop counts anyway (reads longer than the position), at flush time.  The masks were 30 of the
  // strided kernel's 94 VALU instructions per chunk; the kernel is mostly bound by how its three load streams arrive, so the
  // gain is 2.3 % (10M trimmed 150 bp reads 0.5300 -> 0.5178 ms in one process, 0.696 -> 0.712 of peak).
  constexpr bool FAST_FIXED = FIXED && (!SV || NP);   // no tail masks
  // LUTV (round 5; the fused adapter scan on fixed-length reads, 16 positions per lane — the VALU-bound variant): base codes by
  // table lookup (v_perm_b32) instead of three SWAR indicators per dword, see consume
  constexpr bool LUTV = W16 && ADAPT && FIXED && MODE == 0;
  // code table, selector (byte >> 1) & 7: A 000 -> 3, C 001 -> 1, T 010 -> 2, G 011 -> 0, 100 .. 111 (H-O, X-_, N) -> 3
  constexpr uint32_t lut_cc_lo = 0x00020103u, lut_cc_hi = 0x03030303u;
  // the five-bit key (byte & 31) the classes 001 / 010 / 011 must have for that code to be quack.c:150's: C = 3, T = 20, G = 7
  constexpr uint32_t lut_k5 = 0x07140300u;
  // code -> "is T" (2), "is C" (1) as 0 / 1 bytes
  constexpr uint32_t lut_is_t = 0x00010000u, lut_is_c = 0x00000100u;
  constexpr bool UNIFORM = FIXED && !SV;              // every read one length: lengths in closed form
  uint32_t events = 0, steps_v = 0;
  uint32_t fixed_mask = 0;   // FIXED: which of the lane's 8K positions are bases of a read (the same for every row): bit i = position cpos + i

  uint32_t cur_tile = 0xFFFFFFFFu, reads_in_tile = 0;   // (work loop below) the tile whose counters the LDS holds; reads (rows) since its last flush
  auto spill = [&]() {
    if constexpr (FAST_FIXED) {
      // No tail masks (round 5: the short form).  `valid` needs no counter at all — every read of the tile covers every position
      // that is flushed (UNIFORM: rows since the last flush x reads per row; NP: from the lengths) —, T and C share one word
      // (16 bits each: a workgroup sees at most 65,535 reads between two flushes, kMaxReadsPerSlice) and nothing is skipped: two
      // LDS adds per position instead of four under four branches.  A spill was 3.5 us per wave of the adapter kernel and the
      // four waves of a SIMD take turns at the LDS: 10 us at the end of a launch, and again every 255 steps (-DQK_TIMING).
      // Lanes that own no position of a read (feeders, the lanes behind a row) add nothing; positions of a lane that lie behind
      // its read count into columns nobody flushes.
      if (fixed_mask != 0u) {
#pragma unroll
        for (int d = 0; d < 2 * K; ++d) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const uint32_t off = (8u * chk[d >> 1] + 4u * (d & 1) + b) * 4u;
            uint32_t t = (acc_t[d] >> (8 * b)) & 0xFFu;
            uint32_t c = (acc_c[d] >> (8 * b)) & 0xFFu;
            uint32_t g = (acc_g[d] >> (8 * b)) & 0xFFu;
            if constexpr (LUTV) {   // acc_t / acc_c count the T / C bytes themselves, acc_g the bytes that are A-like or C
              g = steps_v - g - t;
            } else {
              t = steps_v - t;
              c = steps_v - c;
              g = steps_v - g;
            }
            lds_add(lds_base, off + 4u * TP, t | (c << 16));
            lds_add(lds_base, off + 12u * TP, g);
          }
        }
      }
#pragma unroll
      for (int d = 0; d < 2 * K; ++d) acc_v[d] = acc_t[d] = acc_c[d] = acc_g[d] = 0;
      since_spill = 0;
      events = 0;
      steps_v = 0;
      return;
    }
#pragma unroll
    for (int d = 0; d < 2 * K; ++d) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t off = (8u * chk[d >> 1] + 4u * (d & 1) + b) * 4u;
        uint32_t v = events - ((acc_v[d] >> (8 * b)) & 0xFFu);   // acc_v counts the events in which the byte was masked
        if (v == 0) continue;  // nothing valid => no T/C/G either
        lds_add(lds_base, off, v);
        // (T and C in one word, 16 bits each — a workgroup sees at most 65,535 reads between two flushes — and no test for zero:
        //  three LDS adds per position instead of four under three more branches, as in the short form above)
        const uint32_t t = events - ((acc_t[d] >> (8 * b)) & 0xFFu);
        const uint32_t c = events - ((acc_c[d] >> (8 * b)) & 0xFFu);
        const uint32_t g = events - ((acc_g[d] >> (8 * b)) & 0xFFu);
        lds_add(lds_base, off + 4u * TP, t | (c << 16));
        lds_add(lds_base, off + 12u * TP, g);
      }
      acc_v[d] = acc_t[d] = acc_c[d] = acc_g[d] = 0;
    }
    since_spill = 0;
    events = 0;
    steps_v = 0;
  };

  auto zero_lds = [&]() {
#if defined(QK_ABL) && (QK_ABL & 2048)   /* experiment: ... without clearing the LDS image */
    return;
#endif
    if constexpr (WIDE) {   // the planes (with the ring's columns: process() sets those) and the letter / length / kmer rows + misc[0]
      for (uint32_t pl = 0; pl < wide_np; ++pl)
        for (uint32_t i = tid; i < 128u * 64u; i += T) lds_raw[pl * kWidePlaneStride + i] = 0;
      for (uint32_t i = tid; i < 6u * TP + 1u; i += T) lds_base[i] = 0;
    } else
    for (uint32_t i = tid; i < hist_words + (ADAPT ? 6u : 5u) * TP + 1u; i += T) lds[i] = 0;
  };

  // ---- flush: LDS -> planar u64 table; zero counters are skipped.  One wave
  // per quality row, lanes along positions (contiguous 512-B atomics).
  auto flush = [&](uint32_t tile) {
#if defined(QK_ABL) && (QK_ABL & 4096)   /* experiment: ... without the flush */
    return;
#endif
    const uint32_t P0 = tile * p.tile_pos;
    if (!UNIFORM && tile == 0 && n_gt10) {
      lds_add(lds_misc, 0, n_gt10);
      n_gt10 = 0;
    }
    __syncthreads();
    QK_MARK(8);
    // fixed-length batches count unmasked: columns at and behind read_len hold the next read's bytes
    uint32_t pos_limit = (FIXED && p.read_len < p.table_len) ? p.read_len : p.table_len;
    if (NP && p.len_limit < pos_limit) pos_limit = p.len_limit;   // (columns behind the longest read hold pads and the next read's bytes)
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    // grouped rows: position pp of the reads lives in the columns pp, pp + gstride, ... (one per read of a row); summed here
    const uint32_t span = (GROUPS && GRP > 1u) ? p.read_len : TP;
    // (every workgroup starting at another row — same-address queueing at the L2 — was measured again in round 4, also on 0.2 ms
    // launches of 36 bp reads where the workgroups do finish in step: nothing, 0.1864 / 0.1847 ms)
    // (round 5: a lane's position owns the same words in every row, so the rows of a wave go innermost — one address per
    //  (read of the row, replica), six independent LDS reads behind it, one wait — where the first form walked row by row with an
    //  address computation and a waited-for read per word: 4.2-6.7 us of every launch by the in-kernel stamps)
    constexpr uint32_t kRowsPerWave = (91u + T / 64u - 1u) / (T / 64u);   // quality rows 33..123 (quack.c:203: bin = byte - 33)
    constexpr uint32_t kRowsAtOnce = FIXED ? 3u : 2u;   // (registers: six rows at once pushed half of the variants into scratch; the ragged kernels, which flush inside their work loop, keep one)
    const uint32_t row_words = WIDE ? 64u : 32u;
    const uint32_t *hist = WIDE ? lds_raw : lds;
#pragma unroll 1
    for (uint32_t k0 = 0; k0 < kRowsPerWave; k0 += kRowsAtOnce) {
#pragma unroll 1
      for (uint32_t pp = lane; pp < span; pp += 64u) {
        uint32_t cnt[kRowsAtOnce];
#pragma unroll
        for (uint32_t k = 0; k < kRowsAtOnce; ++k) cnt[k] = 0;
        for (uint32_t gi = 0; gi < GRP; ++gi) {
          const uint32_t col = pp + gi * GS;
          const uint32_t c8 = col >> 3, j = col & 7u;
          for (uint32_t rep = 0; rep < R; ++rep) {
            const uint32_t hs = rep * 4u + (j & 3u);   // the set
            const uint32_t at = WIDE ? qhist_index_wide(0u, hs * (CH / 2u) + (c8 >> 1), c8 & 1u) : qhist_index(0u, hs * CH + lds_col(c8));
            uint32_t w[kRowsAtOnce];
#pragma unroll
            for (uint32_t k = 0; k < kRowsAtOnce; ++k) {   // (a last row past 123: read row 127, unused)
              const uint32_t row = 33u + wave + (k0 + k) * (T / 64u);
              w[k] = hist[at + (row < 127u ? row : 127u) * row_words];
            }
#pragma unroll
            for (uint32_t k = 0; k < kRowsAtOnce; ++k) cnt[k] += (j >> 2) ? (w[k] >> 16) : (w[k] & 0xFFFFu);
          }
        }
        const uint32_t pos = P0 + pp;
        if (pos < pos_limit) {
#pragma unroll
          for (uint32_t k = 0; k < kRowsAtOnce; ++k) {
            const uint32_t row = 33u + wave + (k0 + k) * (T / 64u);
            if (row <= 123u && cnt[k] != 0) {
              if (p.table32) atomicAdd(&p.table32[(uint64_t)(row - 33u) * TL + pos], cnt[k]);
              else atomicAdd(&p.table[(uint64_t)(row - 33u) * TL + pos], (unsigned long long)cnt[k]);
            }
          }
        }
      }
    }
    QK_MARK(9);
    for (uint32_t pp = tid; pp < span; pp += T) {
      uint32_t v = 0, t = 0, c = 0, g = 0;
      for (uint32_t gi = 0; gi < GRP; ++gi) {
        const uint32_t col = pp + gi * GS;
        {   // (see spill: T | C << 16 in one word; batches without tail masks keep no `valid` row)
          const uint32_t tc = lds_base[TP + col];
          t += tc & 0xFFFFu;
          c += tc >> 16;
          if constexpr (!FAST_FIXED) v += lds_base[col];
        }
        g += lds_base[3u * TP + col];
      }
      if constexpr (UNIFORM) v = reads_in_tile * GRP;   // every read of the tile has every position below read_len
      const uint32_t pos = P0 + pp;
      if (NP) {   // reads (of this workgroup, since its last flush) longer than the position: from the lengths counted in the loop
        v = 0;
        for (uint32_t j = pp; j < TP; ++j) v += lds_len[j];
      }
      if (v == 0 || pos >= pos_limit) continue;
      const uint32_t a = v - t - c - g;                     // content[] order: A,T,C,G (quack.c:150)
      if (p.table32) {
        uint32_t *row0 = &p.table32[(uint64_t)kRowContent * TL + pos];
        if (a) atomicAdd(row0, a);
        if (t) atomicAdd(row0 + TL, t);
        if (c) atomicAdd(row0 + 2u * TL, c);
        if (g) atomicAdd(row0 + 3u * TL, g);
      } else {
        unsigned long long *row0 = &p.table[(uint64_t)kRowContent * TL + pos];
        if (a) atomicAdd(row0, (unsigned long long)a);
        if (t) atomicAdd(row0 + TL, (unsigned long long)t);
        if (c) atomicAdd(row0 + 2u * TL, (unsigned long long)c);
        if (g) atomicAdd(row0 + 3u * TL, (unsigned long long)g);
      }
    }
    QK_MARK(10);
    if (!UNIFORM && !p.lengths_done) {
      // reads that END in this tile (staged once per tile they reach, so each read counts exactly once)
      for (uint32_t pp = tid; pp < TP; pp += T) {
        const uint32_t c = lds_len[pp];
        if (c != 0 && P0 + pp < p.table_len)
          atomicAdd(&p.table[(uint64_t)kRowLength * TL + P0 + pp], (unsigned long long)c);
      }
    }
    if (ADAPT && p.count_in_kernel) {
      for (uint32_t pp = tid; pp < TP; pp += T) {
        const uint32_t c = lds_kmer[pp];
        if (c != 0 && P0 + pp < p.table_len)
          atomicAdd(&p.table[(uint64_t)kRowKmer * TL + P0 + pp], (unsigned long long)c);
      }
    }
    if (tile == 0) {
      if (UNIFORM) {
        if (tid == 0 && fixed_reads != 0) {
          const unsigned long long n = (unsigned long long)fixed_reads * GRP;   // (fixed_reads counts rows)
          if (p.read_len != 0)
            atomicAdd(&p.table[(uint64_t)kRowLength * TL + p.read_len - 1u], n);   // quack.c:219
          if (p.no_adapters && p.read_len > 10u)
            atomicAdd(&p.table[(uint64_t)kRowKmer * TL + 10u], n);                 // quack.c:215-217, i == 10
        }
      } else {
        if (tid == 0 && p.no_adapters && lds_misc[0] != 0)
          atomicAdd(&p.table[(uint64_t)kRowKmer * TL + 10u], (unsigned long long)lds_misc[0]);
      }
    }
    fixed_reads = 0;
  };

  // ---- one work item: reads [r_begin, r_end) x positions of `tile`
  auto process = [&](uint32_t tile, uint64_t r_begin, uint64_t r_end_in) {
    const uint32_t P0 = tile * p.tile_pos;
    // first position of the owned chunk; halo lanes of tile 0 would sit before
    // the read and are parked beyond any read instead (they then never load)
    const int32_t cpos_s = (int32_t)P0 + 8 * K * ch_signed;
    const uint32_t cpos = cpos_s < 0 ? 0xFFFFFF00u : (uint32_t)cpos_s;
    const bool sorted = !FIXED && p.order != nullptr;
    const uint64_t list_len = sorted ? p.reach[tile] : p.n_reads;   // reads this tile has to look at
    uint64_t r_end = r_end_in;
    if (r_end > list_len) r_end = list_len;
    const uint32_t slice_reads = r_end > r_begin ? (uint32_t)(r_end - r_begin) : 0u;
    if (FIXED && tile == 0) fixed_reads += slice_reads;
    // One tile: every read of the slice is this workgroup's alone, so the per-read first-hit words need no
    // pass of their own before the launch and no counting kernel behind it (adapter_count_kernel) —
    // 45 us of a 1.5 ms step on 10M x 300.  Reset here, count below, both with device-coherent
    // accesses: the atomicMin of the drain happens at the memory side.
    if (fh_ring) {
      for (uint32_t i = tid; i < FHW; i += T) *fh_at(i) = kNoHit;
      fh_folded = 0;
      fh_next = fh_group;
      __syncthreads();
    } else if (ADAPT && p.count_in_kernel) {
      for (uint32_t i = tid; i < slice_reads; i += T)
        __hip_atomic_store(&p.first_hit[r_begin + i], kNoHit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (every access to these words is an agent-scope atomic and the barrier orders them: no __threadfence —
      // on this part it writes back and invalidates the XCD's L2 under 31 other streaming workgroups, +12 % kernel time)
      __syncthreads();
    }
    // 32-bit byte offsets relative to a 4-byte-aligned, workgroup-uniform base
    uint64_t slice_base = 0;
    // (sorted: the slice's reads lie anywhere in the batch, which is < 4 GiB then: base 0)
    if (slice_reads && !sorted) slice_base = FIXED ? r_begin * p.stride : p.offsets[r_begin];
    if (!FIXED) {
      // offsets[] comes through a vector load: tell the compiler the value is the
      // same in every lane, or the two base pointers below live in VGPRs and
      // every load of the loop pays two v_readfirstlane + s_nop for them
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)slice_base);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(slice_base >> 32));
      slice_base = ((uint64_t)hi << 32) | lo;
    }
    const uint64_t base_al = slice_base & ~3ull;
    const uint8_t *qbase = p.qual + base_al;
    const uint8_t *sbase = p.seq + base_al;
    const uint64_t *obase = STAGED ? p.offsets + (sorted ? 0 : r_begin) : nullptr;
    const uint32_t *ord = sorted ? p.order + r_begin : nullptr;
    const uint64_t room = p.total_bytes - base_al;   // total_bytes >= base_al when slice_reads > 0
    const uint32_t off_limit = room < 0xFFFFFFF0ull ? (uint32_t)room : 0xFFFFFFF0u;
    const uint32_t cposp = cpos + (uint32_t)(slice_base & 3ull);
    // the lane's chunks cover the same bytes of every read of a fixed-length batch
    if (FIXED) {
      fixed_mask = 0u;
      uint32_t g, pos;
      col_split(cpos, g, pos);
      for (uint32_t i = 0; i < 8u * K; ++i) {
        if (lane_on && cpos + i < row_len && g < GRP && pos < p.read_len) fixed_mask |= 1u << i;
        ++pos;
        if (GROUPS && GRP > 1u && pos == GS) {
          pos = 0u;
          ++g;
        }
      }
    }

    // Strided rows (16 positions per lane): length_count (quack.c:219), the kmers==NULL count (quack.c:215) and the check of
    // device-side lengths[] for the slice's reads in ONE pass in front of the step loop — sixteen coalesced loads in flight per
    // thread, equal lengths of a wave added up first (70 % of trimmed reads have one length).  Round 5's first build took the length
    // with the read's bytes in every step, as the 8-position kernel does: one more load per lane and step, the count under nested
    // exec masks, 98 VALU + 27 SALU per chunk against the padded twin's 82 + 19.
    if constexpr (SV && W16) {
      if (!p.lengths_done && slice_reads) {
        const uint32_t total = slice_reads * GRP;
        const uint32_t *lr = p.lengths + (size_t)r_begin * GRP;
        const uint32_t lim = p.len_limit < TP ? p.len_limit : TP;
        uint32_t mine = 0;
        bool bad = false;
        // (the batch's longest length — what the untrimmed reads have — is counted by one ballot per load and added once per wave;
        //  the other lengths go to the LDS lane by lane unless a wave meets many of them, which the leader loop of wave_count_lds
        //  is for.  The in-kernel stamps had this pass at 22 us per launch of 10M reads with the leader loop on everything.)
        const uint32_t hot = lim;
        uint32_t hot_n = 0;
        for (uint32_t base = 0; base < total; base += 16u * T) {   // (whole workgroup, whole waves: the counting uses ballots)
          uint32_t v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const uint32_t i = base + (uint32_t)j * T + tid;
            v[j] = i < total ? lr[i] : 0u;
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const uint32_t len = v[j];
            bad |= len > lim;
            mine += (len > 10u && len <= lim) ? 1u : 0u;
            const bool is_hot = len == hot && len != 0u;
            hot_n += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(is_hot));
            const bool rest = len != 0u && len < hot;
            if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(rest)) > 24) wave_count_lds(lds_len, len - 1u, rest);
            else if (rest) atomicAdd(&lds_len[len - 1u], 1u);
          }
        }
        if ((tid & 63u) == 0u && hot_n) atomicAdd(&lds_len[hot - 1u], hot_n);
        if (bad) atomicOr(p.status, kStatusBadLength);   // (device-side lengths[] are only seen here)
        n_gt10 += mine;
      }
    }

    // Fixed-length batches: one pass over the slice, read r at r*L.  Ragged
    // batches: passes of stage_reads reads; each pass first stages, in LDS, the
    // descriptors of the reads that reach this tile (coalesced offsets[] read,
    // wave-ballot compaction), so the loop below never waits on an offsets
    // round trip and never spends a slot on a read that ends before the tile.
    // the offsets of the next pass are requested one pass ahead (each thread
    // stages the read `tid` of a pass), so staging never waits on memory
    const uint32_t *lbase = (!STAGED || !p.lengths) ? nullptr : p.lengths + (sorted ? 0 : r_begin);
    // read number `i` of the slice -> index into offsets[] / lengths[] (relative to obase / lbase)
    auto read_id = [&](uint32_t i) { return ord ? ord[i] : i; };
    uint64_t pf0 = 0, pf1 = 0;
    uint32_t pfid = 0;
    if (STAGED && tid < slice_reads) {
      pfid = read_id(tid);
      pf0 = obase[pfid];
      pf1 = lbase ? pf0 + lbase[pfid] : obase[pfid + 1u];
    }
    // first-hit ring -> kmer_count row, for the reads [fh_folded, upto) of the slice (every candidate of theirs has been
    // drained: the callers drain and meet at a barrier first).  quack.c:211-217: i ends one past the first window
    // found; counted iff i < l.
    auto fold_first_hits = [&](uint32_t upto) {
      // (grouped rows: the ring holds one word per READ, row * group + the read's place in the row)
      for (uint32_t rel = fh_folded * GRP + tid; rel < upto * GRP; rel += T) {
        const uint32_t v = *fh_at(rel & FHM);
        if (v == kNoHit) continue;
        if (v + 1u < TP) lds_add(lds_kmer, 4u * (v + 1u), 1u);   // (the drain only records hits that count: v + 1 < length)
        *fh_at(rel & FHM) = kNoHit;
      }
      fh_folded = upto;
    };
    for (uint32_t pass = 0; pass < slice_reads; pass += STAGED ? SR : slice_reads) {
      uint32_t n_list = slice_reads;   // FIXED: every read of the slice
      if (STAGED) {
        const uint32_t nb = slice_reads - pass < SR ? slice_reads - pass : SR;
        static_assert(T == 1024 || T == 512 || T == 256, "staging assumes kStageReads is a multiple of T");
        __syncthreads();               // the previous pass has been consumed
        if (tid == 0) lds_misc[2] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < (nb + 63u) / 64u * 64u; i += T) {   // whole waves: ballot below
          uint64_t o0 = 0;
          uint32_t len = 0, id = 0;
          if (i < nb) {
            if (T == SR) {    // one read per thread: it was prefetched
              o0 = pf0;
              len = (uint32_t)(pf1 - pf0);
              id = pfid;
            } else {
              id = read_id(pass + i);
              o0 = obase[id];
              len = lbase ? lbase[id] : (uint32_t)(obase[id + 1] - o0);
            }
            if ((AL || p.check_aligned) && (o0 & 127u) != 0) atomicOr(p.status, kStatusNotAligned);   // the producer's promise does not hold
          }
          if (!p.lengths_done) {   // wave-uniform
            // length_count (quack.c:219) of the reads that END in this tile — every read is staged
            // once for every tile it reaches, so once for its last — and, at tile 0, the kmers==NULL
            // count (quack.c:215).  Real batches are dominated by a few lengths (untrimmed reads), and
            // 64 lanes adding to one LDS address serialise (143k reads of one length: 1.89 ms instead of
            // 0.68), so the wave first adds up its first-met lengths as a whole.
            if (tile == 0) n_gt10 += (i < nb && len > 10u) ? 1u : 0u;
            const uint32_t lp = len - 1u - P0;   // position of the last base inside this tile (if it is)
            bool todo = i < nb && len > P0 && lp < TP;
            uint64_t left = __builtin_amdgcn_ballot_w64(todo);
            for (int k = 0; k < 4 && left; ++k) {
              const uint32_t leader = (uint32_t)__builtin_ctzll(left);
              const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)lp, (int)leader);
              const uint64_t same = __builtin_amdgcn_ballot_w64(todo && lp == v);
              const uint32_t cnt = (uint32_t)__builtin_popcountll(same);
              if (cnt < 4u) break;   // all different: plain atomics do as well
              if (lane_id == leader) lds_add(lds_len, v * 4u, cnt);
              if (lp == v) todo = false;
              left &= ~same;
            }
            if (todo) lds_add(lds_len, lp * 4u, 1u);
          }
          const bool reach = i < nb && len > P0;
          const uint64_t vote = __builtin_amdgcn_ballot_w64(reach);
          uint32_t wave_base = 0;
          if (lane_id == 0 && vote) wave_base = atomicAdd(&lds_misc[2], (uint32_t)__builtin_popcountll(vote));
          wave_base = __builtin_amdgcn_readfirstlane(wave_base);
          if (reach) {
            const uint32_t slot_i = wave_base + (uint32_t)__builtin_popcountll(vote & ((1ull << lane_id) - 1ull));
            lds_list[slot_i] = make_uint2((uint32_t)(o0 - base_al), len);
            // first_hit[] is indexed by the read, not by the list slot
            if (ADAPT) lds_ridx[slot_i] = sorted ? id : (uint32_t)r_begin + id;
          }
        }
        __syncthreads();
        n_list = lds_misc[2];
        if (T == SR && pass + SR + tid < slice_reads) {
          pfid = read_id(pass + SR + tid);
          pf0 = obase[pfid];
          pf1 = lbase ? pf0 + lbase[pfid] : obase[pfid + 1u];
        }
      }

      // One step = RW*U reads: `issue` computes the addresses and starts the
      // loads, `consume` histograms them.  With PD > 1 the loads of step k+1
      // are in flight while step k is consumed (PD register sets).
      // Ragged: the descriptors of a step are read from the staged list one step
      // AHEAD (load_desc, right after the previous step's loads went out and
      // before its ds_adds), so that the LDS read does not queue behind a
      // step's 32 atomics when the next global loads want to start.
      QK_MARK(2);
      uint2 de[U];
      auto load_desc = [&](uint32_t it) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t rel = it + (uint32_t)u * RW + ri;
          const bool in_list = rel < n_list && ri < RW;
          de[u] = lds_list[in_list ? rel : 0u];
        }
      };
      // FIXED: byte offset of the lane's chunk in the reads of step 0 — the step's own part, it * read_len,
      // is wave-uniform and computed by the scalar unit (one v_add per read instead of a v_mad_u64_u32,
      // the only 32-bit integer multiply-add there is: quarter rate)
      uint32_t fixed_off0[U];
#pragma unroll
      for (int u = 0; u < U; ++u) fixed_off0[u] = FIXED ? ((uint32_t)u * RW + ri) * p.stride + cposp : 0u;
      // strided: lengths[] of the slice's reads (the candidate check reads them)
      const uint32_t *lrow = SV ? p.lengths + (size_t)r_begin * GRP : nullptr;
      auto issue = [&](uint32_t it, LoadT (&q)[U], LoadT (&s)[U], uint32_t (&nv)[U], uint32_t (&sk)[U],
                       uint32_t (&rl)[U]) __attribute__((always_inline)) {
        const uint32_t it_bytes = FIXED ? (uint32_t)__builtin_amdgcn_readfirstlane((int)it) * p.stride : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t rel = it + (uint32_t)u * RW + ri;   // index into the slice (FIXED) / the staged list
          rl[u] = rel;
          const bool in_list = rel < n_list && ri < RW;
          uint32_t off, len;
          if (FIXED) {
            off = it_bytes + fixed_off0[u];
            len = row_len;
          } else {
            const uint2 e = PD > 1 ? lds_list[in_list ? rel : 0u] : de[u];
            off = e.x + cpos;
            len = e.y;
          }
          // bytes of this lane's chunk(s) inside the read; feeder lanes (ADAPT) load and
          // compute codes like their originals but count nothing
          uint32_t n_raw = (in_list && len > cpos) ? len - cpos : 0u;
          n_raw = n_raw > 8u * K ? 8u * K : n_raw;
          nv[u] = lane_on ? n_raw : 0u;
          // strided: the read's own length, in flight together with its bytes (consume turns it into n)
          // (NP: only the lane that counts the length needs it; loading it in that lane alone was measured: 0.5082 / 0.5091 ms)
          // (a 32-bit index from the slice's own, wave-uniform base: the step's part by the scalar unit, the lane's part a constant —
          //  64-bit index arithmetic here was two quarter-rate v_mad_u64_u32 per lane and step; a row that is not in the list loads
          //  word 0 and is told apart in consume)
          // (8 positions per lane — the builds of rounds 2-4, one of them held to 64 VGPRs — keep their load under the exec mask: the
          //  unconditional form measured 8 % slower there, 0.512 -> 0.553 ms per 10M trimmed reads, for no reason the listing shows)
          if (SV) {
            // (16 positions per lane: the lengths of a slice are counted in one pass in front of the step loop — process() — and the
            //  candidate check loads the few it needs; the loop itself carries none)
            if constexpr (W16) nv[u] = in_list ? 1u : 0u;
            else nv[u] = in_list ? p.lengths[(size_t)r_begin + rel] : 0u;
          }
          off = off < off_limit ? off : off_limit;   // stay inside the buffer (+ slack)
          sk[u] = AL ? 0u : (off & 3u);
          off &= (AL && !FIXED) ? (W16 ? ~15u : ~7u) : ~3u;
          if constexpr (W16) {
            // two adjacent chunks, one dwordx4 per array (ragged: 16-byte aligned — the reads start on cache
            // lines; fixed length: dword aligned).  Lanes past the end of their read fetch the slice's first line.
            if (!FIXED) off = n_raw != 0 ? off : 0u;
            q[u] = load16_aligned(qbase + off);
            s[u] = load16_aligned(sbase + off);
          } else if (AL && FIXED) {
            // read_len is a multiple of 4: every chunk starts on a dword
            q[u] = load8_aligned(qbase + off);
            s[u] = load8_aligned(sbase + off);
          } else if (AL) {
            // chunks past the end of their read fetch the slice's first line instead (hot in
            // the cache, and every byte of it is masked anyway): cheaper than giving the
            // registers defaults and loading under an exec mask (6 v_mov + 3 per read)
            off = n_raw != 0 ? off : 0u;
            q[u] = load8_aligned(qbase + off);
            s[u] = load8_aligned(sbase + off);
          } else if (FIXED) {
            q[u] = load12_aligned(qbase + off);
            s[u] = load12_aligned(sbase + off);
          } else {
            // chunks past the end of their read fetch nothing
            q[u] = u32x3{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            s[u] = q[u];
            if (n_raw != 0) {
              q[u] = load12_aligned(qbase + off);
              s[u] = load12_aligned(sbase + off);
            }
          }
        }
      };
      // Exact check of the queued candidates {plo, prev2 | hits << 2 | lane << 10 | rel << 16}, one entry per
      // lane.  Position and length of the chunk come back from the lane that queued it (ds_bpermute)
      // and from the read's descriptor: the queue never outlives its pass.
      // W16: {own32, the previous lane's code word (bits 0-15: the 8 positions before the lane's, 16-17: the one before
      // those), probe bits << 24 | lane << 16, rel} — one entry per LANE, 16 windows.
      auto drain_candidates = [&]() {
        constexpr uint32_t NW = 8u * K;                 // windows per entry
        for (uint32_t i = lane_id; i < ((cand_n + 63u) & ~63u); i += 64u) {   // whole waves: bpermute below
          uint32_t src, rel, hits, s_lo, s_hi;
          if constexpr (W16) {
            const uint4 e = cand_q16[i < cand_n ? i : 0u];
            src = (e.z >> 16) & 63u;
            rel = e.w;
            // eight probe bits -> sixteen windows: probe i passed = windows 2i (the 9-mer is its suffix) and 2i + 1 (its prefix)
            uint32_t x = e.z >> 24;
            x = (x | (x << 4)) & 0x0F0Fu;
            x = (x | (x << 2)) & 0x3333u;
            x = (x | (x << 1)) & 0x5555u;
            hits = x * 3u;
            s_lo = e.x;
            s_hi = e.y & 0x3FFFFu;
          } else {
            const uint2 e = cand_q[i < cand_n ? i : 0u];
            src = (e.y >> 10) & 63u;
            rel = e.y >> 16;
            hits = (e.y >> 2) & 0xFFu;
            s_lo = e.x;
            s_hi = e.y & 3u;
          }
          const uint32_t cp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)cpos);
          uint32_t len, rd;
          if (FIXED) {
            rd = (uint32_t)r_begin + rel;
            len = (SV && !GROUPS) ? lrow[rel] : p.read_len;   // (strided rows of several reads: per read, below)
          } else {
            len = lds_list[rel].y;
            rd = lds_ridx[rel];
          }
          hits = i < cand_n ? hits : 0u;
          // only windows that end inside the read, at e >= 9 (quack.c:206-213), and BEFORE its last base: those of the lane's
          // windows j whose position base + j lies in [9, ln - 1).  (A first hit that ends on the last base is not counted —
          // quack.c:215, i == l — and nothing can follow it, so leaving it out changes no count and the fold needs no length.)
          auto window_mask = [&](int32_t base, uint32_t ln) -> uint32_t {
            int32_t lo = 9 - base, hi = (int32_t)ln - 1 - base;
            lo = lo < 0 ? 0 : lo;
            hi = hi > (int32_t)NW ? (int32_t)NW : hi;
            return hi > lo ? (((1u << (uint32_t)(hi - lo)) - 1u) << (uint32_t)lo) : 0u;
          };
          const uint64_t stream = ((uint64_t)s_hi << 32) | s_lo;
          // the first window of `h` whose 10-mer is in the table (later windows are later positions), or NW
          auto first_in_table = [&](uint32_t h) -> uint32_t {
            while (h) {
              const uint32_t j = (uint32_t)__builtin_ctz(h);
              h &= h - 1u;
              // the window ending at owned position j = bits [2*(NW-1-j), 2*(NW-1-j)+20) of the (complemented) code stream
              const uint32_t km = ((uint32_t)(stream >> (2u * (NW - 1u - j))) & kKmerMask) ^ kKmerMask;
              const bool in_table = p.bucket_log2 ? bucket_has(lds_buckets, km, p.bucket_mul, p.bucket_log2)
                                                  : ((p.kmer_bits[km >> 5] >> (km & 31u)) & 1u) != 0;
              if (in_table) return j;
            }
            return NW;
          };
          // grouped rows: the lane's positions belong to read g0 of its row from position pos0 on and, past that read's
          // stride, to read g0 + 1 from its position 0 on (a stride is at least 16 positions: two reads at most)
          uint32_t g0, pos0;
          col_split(cp, g0, pos0);
          uint32_t found, ring = rel;   // (ring: the read's word in the first-hit ring)
          if constexpr (SV && GROUPS) {
            // Strided rows: the lengths the check needs come from memory — a twentieth of the lanes ever gets here.  Both are
            // requested first and looked at last: the look-ups need only the lower end of a read's windows (e >= 9), and since
            // it is the FIRST window in the table that counts, the upper end (e <= l - 2) is one comparison behind them.
            // (Loaded where they were used, every check waited for memory — with vmcnt(0), i.e. for the next step's bytes as
            // well: 3 % of the kernel; carried in the queue entry by DPP from a lane that had loaded them: the same.)
            const uint32_t gi0 = g0 < GRP ? g0 : GRP - 1u, gi1 = gi0 + 1u < GRP ? gi0 + 1u : gi0;
            const uint32_t la = lrow[rel * GRP + gi0], lb = lrow[rel * GRP + gi1];
            auto from_nine = [&](int32_t base) -> uint32_t {
              int32_t lo = 9 - base;
              lo = lo < 0 ? 0 : lo;
              return lo >= (int32_t)NW ? 0u : ((((1u << NW) - 1u) >> (uint32_t)lo) << (uint32_t)lo);
            };
            const uint32_t j0 = first_in_table(hits & from_nine((int32_t)pos0));
            uint32_t j1 = NW;
            const bool second = GRP > 1u && g0 + 1u < GRP && pos0 + NW > GS;
            if (second) j1 = first_in_table(hits & from_nine((int32_t)pos0 - (int32_t)GS));
            found = (j0 < NW && pos0 + j0 + 1u < la) ? pos0 + j0 : kNoHit;
            ring = rel * GRP + g0;
            if (GRP > 1u) rd = ring;           // (what tells two reads apart below)
            if (second && j1 < NW && pos0 + j1 - GS + 1u < lb)
              __hip_atomic_fetch_min(fh_at((ring + 1u) & FHM), pos0 + j1 - GS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
          const uint32_t j0 = first_in_table(hits & window_mask((int32_t)pos0, len));
          found = j0 < NW ? pos0 + j0 : kNoHit;
          if (GROUPS && GRP > 1u) {
            ring = rel * GRP + g0;
            rd = ring;           // (what tells two reads apart below)
            if (g0 + 1u < GRP && pos0 + NW > GS) {
              const uint32_t j1 = first_in_table(hits & window_mask((int32_t)pos0 - (int32_t)GS, len));
              if (j1 < NW) __hip_atomic_fetch_min(fh_at((ring + 1u) & FHM), pos0 + j1 - GS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
          }
          // An adapter covers several chunks of its read, and their entries sit next to each other in
          // the queue in ascending position (lane order of one step): only the first of a run of
          // confirmed entries of one read goes to memory (13.5M -> ~3M atomics per 10M spliced reads)
          const uint32_t prev_rd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane_id + 63u) & 63u) << 2), (int)rd);
          const uint32_t prev_found = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane_id + 63u) & 63u) << 2), (int)found);
          const bool covered = lane_id != 0 && prev_rd == rd && prev_found <= found;
#if defined(QK_ABL) && (QK_ABL & 64)   /* experiment: everything but the global atomic */
          if (found != kNoHit && !covered) keep ^= found + rd;
#else
          if (found != kNoHit && !covered) {
            if (fh_ring) __hip_atomic_fetch_min(fh_at(ring & FHM), found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicMin(&p.first_hit[rd], found);
          }
#endif
        }
        cand_n = 0;
      };
      auto consume = [&](uint32_t it_step, const LoadT (&q)[U], const LoadT (&s)[U], const uint32_t (&nv)[U], const uint32_t (&sk)[U],
                         const uint32_t (&rl)[U]) __attribute__((always_inline)) {
      // ADAPT builds run the U reads of a step in two rounds: first the letters of every
      // read (indicators, counters, codes, the four filter probes), then the quality
      // histogram and the probes' answers — so that the LDS reads of the probes are
      // issued ahead of the step's 8*U ds_adds and have long returned when they are
      // looked at (the kernel issues VALU work back to back on four waves per SIMD and
      // cannot afford to wait for an LDS round trip per read).
      uint32_t qwU[U][K][2], ploU[U][K], bytU[U][K][4], nU[U], prevU[U];
      bool liveU[U];
      // The queue is checked at the START of a step (round 3), not behind the push that filled it: the check ends in
      // global atomics, which count against vmcnt like the loads do — gfx9 has one counter — and the next thing the loop
      // does is wait for the loads of the coming step with vmcnt(0).  Behind the push that wait took the atomics' round
      // trip as well (measured by leaving the atomic out: 5 % of the config-3 kernel); here they have a whole step.
#if !(defined(QK_ABL) && (QK_ABL & 32))
      if ((kScalarLoop || fh_ring) && it_step == fh_next) {   // (kScalarLoop: without the ring no step has fh_next's index)
        fh_next += fh_group;
        // a fold point of the first-hit ring (kFhRing): every wave empties its queue, the workgroup meets, and the
        // entries of the reads before this step go to the kmer_count row.  (The reads of THIS group write other slots
        // — a group is at most half the ring —, and the next group starts behind the next barrier.)
#if defined(QK_ABL) && (QK_ABL & 128)   /* experiment: fold points without the forced check of the queues (wrong counts) */
        __syncthreads();
        fold_first_hits(it_step);
#elif defined(QK_ABL) && (QK_ABL & 256)   /* experiment: the forced checks without barrier and fold */
        if (cand_n) drain_candidates();
#elif defined(QK_ABL) && (QK_ABL & 512)   /* experiment: no fold points at all */
#else
        if (cand_n) drain_candidates();
        __syncthreads();
        fold_first_hits(it_step);
#endif
      } else if (ADAPT && MODE == 0 && !W16 && cand_n > kCandCap - 64u) {   // (W16: when a step's entries would not fit, see the push)
        drain_candidates();
      }
#endif
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // bytes past the end of the read -> 0xFF: quality row 127 is discarded
        // at flush time, and 0xFF & 31 matches none of T/C/G.
        uint32_t nl = nv[u];     // valid bytes of the lane's K chunks together
        uint32_t lenv = 0;       // SV: the length that arrived with the bytes (0: the row is not in the list)
        if (SV) {   // nv[u] is the length of the read
          const bool row_in = rl[u] < n_list && ri < RW;
          lenv = (!W16 || row_in) ? nv[u] : 0u;
          nl = (lane_on && lenv > cpos) ? lenv - cpos : 0u;
          nl = nl > 8u ? 8u : nl;
          if (NP) nl = (lane_on && lenv != 0u) ? 8u : 0u;   // (every byte of an existing read row counts: pads are neutral)
          // (16 positions per lane: a row may hold several reads, one of them empty — whether the row exists says the list)
          if (W16) nl = (lane_on && row_in) ? 8u * K : 0u;
          // length_count (quack.c:219) and the kmers==NULL count (quack.c:215) by the lane that owns the read's first
          // chunk: the length is in its register anyway.  (Late round 3; round 2 counted behind the step loop, a pass
          // of its own over lengths[], and gained nothing over the separate kernel: 24 us per 10M reads.)
          if (!W16 && !p.lengths_done && lane_on && chl == 0u && lenv != 0u) {
            const uint32_t len = lenv;
            if (len > p.len_limit || len > TP) atomicOr(p.status, kStatusBadLength);   // (device-side lengths[] are only seen here)
            else {
              lds_add(lds_len, (len - 1u) * 4u, 1u);
              n_gt10 += len > 10u ? 1u : 0u;
            }
          }
        }
        nU[u] = nl;
        liveU[u] = true;
        // nothing of these reads reaches this tile (ragged batches, long
        // reads): the whole wave moves on.  ADAPT needs every lane's codes, but
        // then no lane has a valid window either.
        if (!FIXED && __builtin_amdgcn_ballot_w64(nl != 0) == 0) {
          liveU[u] = false;
          continue;
        }
        if (FAST_FIXED && !ADAPT && nl == 0) continue;   // per lane: past the end of the slice (its last step only)
        uint32_t own16K[K];
        const bool count_me = !FAST_FIXED || nl != 0;   // ragged: every lane (masked bytes take care of themselves)
        // LUTV (round 5): the complemented 2-bit code of every base byte (A and everything else 3, T 2, C 1, G 0) straight
        // from the byte — bits 3..1 of A, C, G, T, N (either case) are 000, 001, 011, 010, 111, so (byte >> 1) & 7 is the
        // selector of a v_perm_b32 whose two source registers are an 8-entry table: four bases per instruction.  Those three
        // bits decide every byte's code correctly EXCEPT the other members of the classes 001 / 010 / 011 (B R S, D E U,
        // F V W and the non-letters that share their low five bits), which quack.c:150 maps to A: a second table holds the
        // five-bit key a class must have (3, 20, 7; 0 = any), and v_msad_u8 — the sum of absolute differences over the bytes
        // whose reference is not 0 — adds up the violations of a whole step in one register.  A wave that meets one (IUPAC
        // codes; real reads hold A C G T N) recomputes the step's codes from the exact "not T / not C / not G" indicators.
        // What it replaces: three indicators per dword (13 VALU) + three v_dot4 per dword to pack them; now 6 + 1.
        uint32_t ccU[2 * K] = {};
        if constexpr (LUTV) {
          const uint32_t sd[4] = {s[u].x, s[u].y, s[u].z, s[u].w};
          uint32_t bad = 0u;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const uint32_t key = (sd[d] >> 1) & 0x07070707u;
            ccU[d] = __builtin_amdgcn_perm(lut_cc_hi, lut_cc_lo, key);
            bad = __builtin_amdgcn_msad_u8(sd[d] & 0x1F1F1F1Fu, __builtin_amdgcn_perm(0u, lut_k5, key), bad);
          }
          if (__builtin_amdgcn_ballot_w64(bad != 0u) != 0) {   // (wave-uniform, rare)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const uint32_t t1 = swar_ne(sd[d], kKeyT), c1 = swar_ne(sd[d], kKeyC), g1 = swar_ne(sd[d], kKeyG);
              ccU[d] = t1 + 2u * c1 + 3u * g1 - 0x03030303u;   // (per byte 0..3: no carries)
            }
          }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
        const uint32_t n = K == 1 ? nl : (nl > 8u * (uint32_t)k ? (nl - 8u * (uint32_t)k > 8u ? 8u : nl - 8u * (uint32_t)k) : 0u);
        uint2 qa, sa;
        if constexpr (W16) {
          qa = k ? make_uint2(q[u].z, q[u].w) : make_uint2(q[u].x, q[u].y);
          sa = k ? make_uint2(s[u].z, s[u].w) : make_uint2(s[u].x, s[u].y);
        } else if constexpr (AL) {
          qa = make_uint2(q[u].x, q[u].y);
          sa = make_uint2(s[u].x, s[u].y);
        } else {
          qa = window8(q[u], sk[u]);
          sa = window8(s[u], sk[u]);
        }
        // Ragged: bytes past the end of the read -> 0xFF (quality row 127 is
        // discarded at flush time, 0xFF & 31 matches none of T/C/G).  Fixed
        // length: no masks at all — the bytes behind a read's last base belong
        // to the next read and count into columns >= read_len, which the flush
        // never looks at; lanes with nothing to count (past the end of the
        // slice, feeder and halo lanes) sit the counting out under the exec mask.
        // (a ragged wave whose lanes all hold 8 valid bytes — the inside of long
        // reads — skips the mask arithmetic: a wave-uniform branch)
        // Everything the tail masks cost sits in ONE wave-uniform branch: the masks, their ORs into
        // the bytes and the count of masked events.  Waves whose lanes all hold 8 valid bytes — the
        // inside of long reads — skip it.
        uint32_t m0 = 0u, m1 = 0u;
        uint32_t qw[2] = {qa.x, qa.y}, sw[2] = {sa.x, sa.y};
        if (!FAST_FIXED && __builtin_amdgcn_ballot_w64(n != 8u) != 0) {
          m0 = n >= 4u ? 0u : (0xFFFFFFFFu << (8u * n));
          m1 = n >= 8u ? 0u : (n <= 4u ? 0xFFFFFFFFu : (0xFFFFFFFFu << (8u * (n - 4u))));
          qw[0] |= m0;
          qw[1] |= m1;
          sw[0] |= m0;
          sw[1] |= m1;
          if (MODE == 0 || MODE == 3) {   // (every lane, also with n == 0: all of its bytes are masked in this event)
            acc_v[2 * k] += m0 & 0x01010101u;   // events in which the byte was masked
            acc_v[2 * k + 1] += m1 & 0x01010101u;
          }
        }
        const uint32_t mk[2] = {m0, m1};
        if (MODE == 1) {
          keep ^= qw[0] ^ qw[1] ^ sw[0] ^ sw[1];
          continue;
        }
        if (ADAPT) {
          qwU[u][k][0] = qw[0];
          qwU[u][k][1] = qw[1];
        } else if ((MODE == 0 || MODE == 2) && count_me) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const uint32_t wr = __builtin_amdgcn_alignbit(qw[jj], qw[jj], rot8);
            qhist_add<0, kHistBase>(wr, mask7, qcol[k][0], jj ? one_hi : one_lo);
            qhist_add<1, kHistBase>(wr, mask7, qcol[k][1], jj ? one_hi : one_lo);
            qhist_add<2, kHistBase>(wr, mask7, qcol[k][2], jj ? one_hi : one_lo);
            qhist_add<3, kHistBase>(wr, mask7, qcol[k][3], jj ? one_hi : one_lo);
          }
        }
        if constexpr (LUTV) {
          // counters from the code bytes: T and C through one table lookup each (the selector is the code), "A or C" is the
          // code's low bit; the spill takes G = events - (A or C) - T.  Codes: one v_dot4 per dword (weights 64 / 16 / 4 / 1).
          if (count_me) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const uint32_t c = ccU[2 * k + d];
              acc_t[2 * k + d] += __builtin_amdgcn_perm(0u, lut_is_t, c);
              acc_c[2 * k + d] += __builtin_amdgcn_perm(0u, lut_is_c, c);
              acc_g[2 * k + d] += c & 0x01010101u;
            }
          }
          constexpr uint32_t kW = 0x01041040u;
          own16K[k] = __builtin_amdgcn_udot4(ccU[2 * k + 1], kW, __builtin_amdgcn_udot4(ccU[2 * k], kW, 0u, false) << 8, false);
        } else
        if (MODE == 0 || MODE == 3) {
          // letter indicators: the adapter scan needs those of the real bytes
          // (feeder lanes and chunk tails must still yield the real codes; window
          // validity is enforced when the candidates are checked); the counters of a
          // ragged batch need every masked byte to read "not equal"
          uint32_t nt[2], nc[2], ng[2];
          const uint32_t raw[2] = {sa.x, sa.y};
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const uint32_t src = ADAPT ? raw[d] : sw[d];
            nt[d] = swar_ne(src, kKeyT);
            nc[d] = swar_ne(src, kKeyC);
            ng[d] = swar_ne(src, kKeyG);
          }
          if (count_me) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const uint32_t inv01 = (ADAPT && !FAST_FIXED) ? (mk[d] & 0x01010101u) : 0u;
              acc_t[2 * k + d] += nt[d] | inv01;
              acc_c[2 * k + d] += nc[d] | inv01;
              acc_g[2 * k + d] += ng[d] | inv01;
            }
          }
          if (ADAPT) {
            // COMPLEMENTED 2-bit codes (A and everything else 3, T 2, C 1, G 0) of the 8
            // owned bases, first base most significant.  The code is LINEAR in the
            // three indicators, nt + 2 nc + 3 ng - 3, so packing four byte codes
            // into eight bits (weights 64, 16, 4, 1) is three v_dot4_u32_u8 — no
            // shifts, no multiplies (v_mul_lo_u32 is quarter rate).
            constexpr uint32_t kW1 = 0x01041040u, kW2 = 0x02082080u, kW3 = 0x030C30C0u, kBias = 0u - 255u;
            const uint32_t c80 = __builtin_amdgcn_udot4(ng[0], kW3, __builtin_amdgcn_udot4(nc[0], kW2, __builtin_amdgcn_udot4(nt[0], kW1, kBias, false), false), false);
            own16K[k] = __builtin_amdgcn_udot4(ng[1], kW3, __builtin_amdgcn_udot4(nc[1], kW2, __builtin_amdgcn_udot4(nt[1], kW1, (c80 << 8) + kBias, false), false), false);
#if defined(QK_ABL) && (QK_ABL & 4)   /* experiment: no codes at all */
            own16K[k] = 0u;
#endif
          }
        }
        }   // k
        if (MODE == 0 || MODE == 3) {
          if (FAST_FIXED && count_me) steps_v += 1u;
          if (!FAST_FIXED) events += 1u;
          if (ADAPT) {
            // W16: what travels from the previous lane is its whole second code word — first chunk in the upper half,
            // second chunk in the lower: the lower half is the 8 positions in front of this lane, bits 16-17 the position
            // before those (a queue entry wants both, and used to fetch the second one with a DPP of its own)
            if constexpr (W16) prevU[u] = from_prev_lane((own16K[0] << 16) | own16K[K - 1]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
              // the 8 positions in front of the chunk: the previous lane's last chunk (DPP), or this lane's own
              // chunk before it
              const uint32_t prev16 = k == 0 ? (W16 ? prevU[u] : from_prev_lane(own16K[K - 1])) : own16K[k - 1];
              const uint32_t plo = (prev16 << 16) | own16K[k];
              ploU[u][k] = plo;
              // The 9-mer that ends at owned position j is bits [2*(7-j), 2*(7-j)+18)
              // of plo.  It is the suffix of the window ending at j and the prefix of
              // the window ending at j+1, and every window has exactly one such 9-mer
              // ending on an even position: the four probes j = 0,2,4,6 cover the
              // chunk's eight windows.  (The filter holds both 9-mers of every adapter
              // 10-mer; it sits at LDS byte 0, so the key field is the address.)
#pragma unroll
#if defined(QK_ABL) && (QK_ABL & 6)   /* experiment: no filter probes */
              for (int m = 0; m < 4; ++m) bytU[u][k][m] = 0u;
              if (QK_ABL & 2) keep ^= plo;   /* (the codes stay alive) */
#else
              for (int m = 0; m < 4; ++m)   // (wide: the filter sits at byte 32768 — the constant goes into the instruction's offset field)
                bytU[u][k][m] = lds_abs_u32(((plo >> (2 * (7 - 2 * m) + 3)) & ((1u << (kFusedFilterLog2 - 3)) - 4u)) + (WIDE ? kWideFilterOff * 4u : 0u));
#endif
            }
          }
        }
            }
      if (ADAPT && MODE == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!liveU[u]) continue;   // (wave-uniform)
          if (!FAST_FIXED || nU[u] != 0) {
#ifdef QK_DUMMY_VALU   // sensitivity probe (tools only): N extra VALU instructions per chunk
#pragma unroll
            for (int q = 0; q < QK_DUMMY_VALU; ++q) {
#if QK_DUMMY_SLOW == 2   // an LDS atomic on quality row 0 (never flushed), same bank pattern as the real ones
              asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(qcol[0][q & 3]), "v"(one_lo), "i"(kHistBase) : "memory");
#elif QK_DUMMY_SLOW
              asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(keep));
#else
              asm volatile("v_xor_b32 %0, %0, %1" : "+v"(keep) : "v"(mask7));
#endif
            }
#endif
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const uint32_t wr = __builtin_amdgcn_alignbit(qwU[u][k][jj], qwU[u][k][jj], rot8);
#if defined(QK_ABL) && (QK_ABL & 1)   /* experiment: no quality atomics (addresses still computed) */
              keep ^= ((wr & 0x7Fu) << 7 | qcol[k][0]) ^ (((wr >> 1) & mask7) | qcol[k][1]) ^ (((wr >> 9) & mask7) | qcol[k][2]) ^ (((wr >> 17) & mask7) | qcol[k][3]);
              continue;
#endif
              if constexpr (WIDE) {   // (the address registers of the lane's first chunk; the second chunk's counters are 32 dwords on)
                if (k == 0) {
                  qhist_add_wide<0, 0u>(wr, mask7, qcol[0][0], jj ? one_hi : one_lo);
                  qhist_add_wide<1, 0u>(wr, mask7, qcol[0][1], jj ? one_hi : one_lo);
                  qhist_add_wide<2, 0u>(wr, mask7, qcol[0][2], jj ? one_hi : one_lo);
                  qhist_add_wide<3, 0u>(wr, mask7, qcol[0][3], jj ? one_hi : one_lo);
                } else {
                  qhist_add_wide<0, 128u>(wr, mask7, qcol[0][0], jj ? one_hi : one_lo);
                  qhist_add_wide<1, 128u>(wr, mask7, qcol[0][1], jj ? one_hi : one_lo);
                  qhist_add_wide<2, 128u>(wr, mask7, qcol[0][2], jj ? one_hi : one_lo);
                  qhist_add_wide<3, 128u>(wr, mask7, qcol[0][3], jj ? one_hi : one_lo);
                }
                continue;
              }
              qhist_add<0, kHistBase>(wr, mask7, qcol[k][0], jj ? one_hi : one_lo);
              qhist_add<1, kHistBase>(wr, mask7, qcol[k][1], jj ? one_hi : one_lo);
              qhist_add<2, kHistBase>(wr, mask7, qcol[k][2], jj ? one_hi : one_lo);
              qhist_add<3, kHistBase>(wr, mask7, qcol[k][3], jj ? one_hi : one_lo);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!liveU[u]) continue;
          const uint32_t n = nU[u];
          uint32_t hits = 0;
          if constexpr (W16) {
            // One answer bit per probe, collected in a shift register: the probe's bit index is the low five bits of a
            // BYTE of the code word once that is shifted by 6 or by 2 (9-mers end 4 bits apart: 14 and 6 become 8 and 0,
            // 10 and 2 likewise), so the SDWA form of v_lshrrev takes its shift amount straight from that byte, and
            // v_alignbit moves bit 0 of the result into the register — two instructions per probe and two shifted copies
            // per word, against shift + bfe + and + or: 21 VALU instructions per step instead of 32.  The queue entry
            // carries the eight probe bits (bits 24..31, probe 4k + m at 24 + 4k + m); the drain spreads them out.
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const uint32_t wa = ploU[u][k] >> 6, wb = ploU[u][k] >> 2;
              hits = __builtin_amdgcn_alignbit(shr_by_byte<1>(bytU[u][k][0], wa), hits, 1);
              hits = __builtin_amdgcn_alignbit(shr_by_byte<1>(bytU[u][k][1], wb), hits, 1);
              hits = __builtin_amdgcn_alignbit(shr_by_byte<0>(bytU[u][k][2], wa), hits, 1);
              hits = __builtin_amdgcn_alignbit(shr_by_byte<0>(bytU[u][k][3], wb), hits, 1);
            }
          } else
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const uint32_t plo = ploU[u][k];
            uint32_t t9[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)   // 0 or ~0
              t9[m] = (uint32_t)__builtin_amdgcn_sbfe((int)bytU[u][k][m], plo >> (2 * (7 - 2 * m)), 1);   // (offset = the operand's low five bits)
            // windows j (suffix) and j+1 (prefix) of every 9-mer that passed
            const uint32_t h8 = (t9[0] & 0x03u) | (t9[1] & 0x0Cu) | (t9[2] & 0x30u) | (t9[3] & 0xC0u);
            hits |= h8 << (8 * k);
          }
          // lanes that count nothing (feeders, halo, past the end) have no windows
          hits = n ? hits : 0u;
#if defined(QK_ABL) && (QK_ABL & 6)
          hits = 0u;
#endif
#if defined(QK_ABL) && (QK_ABL & 8)    /* experiment: probes issued, answers unused */
          keep ^= bytU[u][0][0] ^ bytU[u][0][1] ^ bytU[u][0][2] ^ bytU[u][0][3] ^ bytU[u][K - 1][0] ^ bytU[u][K - 1][1] ^ bytU[u][K - 1][2] ^ bytU[u][K - 1][3];
          hits = 0u;
#endif
#if defined(QK_ABL) && (QK_ABL & 16)   /* experiment: probes and answers, nobody queues */
          keep ^= hits;
          hits = 0u;
#endif
          const uint64_t pushers = __builtin_amdgcn_ballot_w64(hits != 0u);
          if (pushers) {   // (wave-uniform)
            // position cpos-9: the last base of the chunk two chunks back — the previous lane's (W16: its first chunk)
            uint32_t prev2 = 0;
            if constexpr (!W16) prev2 = from_prev_lane(ploU[u][0] >> 16) & 3u;
            if constexpr (W16) {
              // (round 4) the queue is checked when the step's entries would not fit any more (kCandCap16)
#if defined(QK_ABL) && (QK_ABL & 32)
              if (cand_n + (uint32_t)__builtin_popcountll(pushers) > kCandCap16) cand_n = 0;
#else
              if (cand_n + (uint32_t)__builtin_popcountll(pushers) > kCandCap16) drain_candidates();
#endif
            }
            if (hits) {
              const uint32_t at = cand_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(pushers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pushers, 0u));
              if constexpr (W16)   // (the words as they stand: own 16 codes; the previous lane's word, of which the drain reads bits 0-17)
                cand_q16[at] = make_uint4(ploU[u][1], prevU[u], hits | (lane_id << 16), rl[u]);
              else
                cand_q[at] = make_uint2(ploU[u][0], prev2 | (hits << 2) | lane10 | (rl[u] << 16));
            }
            cand_n += (uint32_t)__builtin_popcountll(pushers);
#if defined(QK_ABL) && (QK_ABL & 32)   /* experiment: candidates queued, never checked */
            if (!W16 && cand_n > kCandCap - 64u) cand_n = 0;
#else
            // (room for the next read's entries; behind the last read of a step the check at the start of the next one does)
            if (!W16 && u + 1 < U && cand_n > kCandCap - 64u) drain_candidates();
#endif
          }
        }
      }
        if (MODE == 0 || MODE == 3) {
          since_spill += U;
          if (since_spill + U > 255u) spill();   // byte counters hold <= 255
        }
        // Issue priority by turns (round 5).  The four waves a SIMD holds do not advance alike: the arbiter takes the oldest ready
        // wave first, and in-kernel time stamps (-DQK_TIMING, profiles/r05_phase_timing.log) showed the waves of a workgroup leaving
        // the step loop in the order of their age, 10 us apart per age on config 2 (30 us from first to last: a tail in which every
        // SIMD runs 3, 2, then 1 wave) and 1.5 us apart per fold interval with the adapter scan, where every fold point is a barrier
        // that waits for the youngest.  s_setprio evens that out.  Without the adapter scan: the four priorities go round the
        // four ages every 32 steps (config 2 -1.5 %, trimmed reads -3 %; a turn per step gains nothing there).  With it: the
        // younger a wave, the larger its share of steps at priority 1 — age a in a of every 4 — which takes the skew between
        // two fold points from 4.6 to 1.9 us and config 3 / 150 bp + adapters down 4-5 %.  (Measured in one process against
        // the same build without it; other schedules — one-hot, two of four, the reverse order, longer periods — did less.)
        // (ragged batches — passes of staged reads with a barrier between them — take the second schedule too: 10M packed reads of
        //  120-150 bases 0.612 -> 0.583 ms; the long-read kernel of config 5 measures the same either way — and so do the adapter
        //  kernels of 8 positions per lane: packed 150 bp reads + adapters 0.698 -> 0.649)
        if constexpr (MODE == 0) {
          const uint32_t age = threadIdx.x >> 8;   // waves w, w + 4, ... share a SIMD, in this order of age
          if constexpr (ADAPT || !FIXED) {
            if (age + (prio_step & 3u) >= 4u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
          } else if ((prio_step & 31u) == 0u) {
            switch (((prio_step >> 5) + age) & 3u) {
              case 0: __builtin_amdgcn_s_setprio(0); break;
              case 1: __builtin_amdgcn_s_setprio(1); break;
              case 2: __builtin_amdgcn_s_setprio(2); break;
              default: __builtin_amdgcn_s_setprio(3); break;
            }
          }
          ++prio_step;
        }
      };
      if constexpr (PD > 1) {
        // PD register sets: the loads of the next PD-1 steps are in flight
        // while one step is consumed
        LoadT q[PD][U], s[PD][U];
        uint32_t nv[PD][U], sk[PD][U], rl[PD][U];
#pragma unroll
        for (int d = 0; d < PD - 1; ++d) issue((uint32_t)d * RW * U, q[d], s[d], nv[d], sk[d], rl[d]);
        if constexpr (kScalarLoop) {
          // The VALU-bound variant (fixed-length reads with the adapter scan, 16 positions per lane) gets its loop
          // bookkeeping off the VALU: the list length comes out of a ticket, so the compiler kept the step counter in a
          // VGPR (compare, add, v_readfirstlane per step) — told to be uniform, it lives in an SGPR; and it kept 2 * RW * U
          // in a register and RE-READ RW from the kernel arguments in every step, with an s_waitcnt lgkmcnt(0) that also
          // waits for the previous step's LDS atomics — one opaque increment serves both counters.  -1.4 % on config 3
          // measured inside one process (tools/ab_inproc.py); the other variants showed nothing and keep the plain loop
          // (some spill more with this one).
          const uint32_t n_steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_list);
          uint32_t step_reads = RW * (uint32_t)U;
          asm volatile("" : "+s"(step_reads));
          uint32_t it = 0, ahead = (uint32_t)(PD - 1) * step_reads;
          while (it < n_steps) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
              const int nx = (d + PD - 1) % PD;
              issue(ahead, q[nx], s[nx], nv[nx], sk[nx], rl[nx]);
              ahead += step_reads;
              if (d == 0 || it < n_steps) consume(it, q[d], s[d], nv[d], sk[d], rl[d]);
              it += step_reads;
            }
          }
        } else
        for (uint32_t it = 0; it < n_list; it += (uint32_t)PD * RW * U) {
#pragma unroll
          for (int d = 0; d < PD; ++d) {
            const int nx = (d + PD - 1) % PD;
            issue(it + (uint32_t)(d + PD - 1) * RW * U, q[nx], s[nx], nv[nx], sk[nx], rl[nx]);
            if (d == 0 || it + (uint32_t)d * RW * U < n_list) consume(it + (uint32_t)d * RW * U, q[d], s[d], nv[d], sk[d], rl[d]);
          }
        }
      } else {
        if (STAGED) load_desc(0u);
        for (uint32_t it = 0; it < n_list; it += RW * U) {
          LoadT q[U], s[U];
          uint32_t nv[U], sk[U], rl[U];
          issue(it, q, s, nv, sk, rl);
          if (STAGED) load_desc(it + RW * U);
          consume(it, q, s, nv, sk, rl);
        }
      }
#if !(defined(QK_ABL) && (QK_ABL & 32))
      if (ADAPT && cand_n) drain_candidates();   // the entries refer to this pass's read list
#endif
    }
    QK_MARK(3);
#ifdef QK_TIMING
    if ((threadIdx.x & 63u) == 0 && blockIdx.x < 1024u) qk_wtime[blockIdx.x * 16u + (threadIdx.x >> 6)] = wall_clock64();
#endif
    // (the letter counters go to the LDS before the workgroup meets: a wave that left the loop early spills while the others still run)
    if (MODE == 0 || MODE == 3) spill();
    if (fh_ring) {
      __syncthreads();   // every wave has drained its queue (above)
      fold_first_hits(slice_reads);
    } else if (ADAPT && p.count_in_kernel) {
      // quack.c:211-217: i ends one past the first window found; counted iff i < l
      __syncthreads();
      // (eight loads in flight per thread: they bypass the caches, and one round trip per read — 38 in a row
      // for a slice of 39k reads — cost 0.1 ms at the end of every workgroup)
      for (uint32_t i0 = tid; i0 < slice_reads; i0 += 8u * T) {
        uint32_t fh[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t i = i0 + (uint32_t)k * T;
          fh[k] = i < slice_reads ? __hip_atomic_load(&p.first_hit[r_begin + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kNoHit;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (fh[k] == kNoHit) continue;
          const uint64_t r = r_begin + i0 + (uint32_t)k * T;
          const uint32_t len = (FIXED && !SV) ? p.read_len : (p.lengths ? p.lengths[r] : (uint32_t)(p.offsets[r + 1] - p.offsets[r]));
          if (fh[k] + 1u < len && fh[k] + 1u < TP) lds_add(lds_kmer, 4u * (fh[k] + 1u), 1u);
        }
      }
    }
    QK_MARK(4);
    return slice_reads;
  };

  // ---- work loop
  auto tile_reads = [&](uint32_t tile) -> uint64_t { return (!FIXED && p.reach) ? p.reach[tile] : p.n_reads; };
  auto run_item = [&](uint32_t tile, uint64_t rb, uint64_t re_in) {
    const uint64_t nt = tile_reads(tile);
    const uint64_t re = re_in < nt ? re_in : nt;
    const uint32_t upcoming = re > rb ? (uint32_t)(re - rb) : 0u;
    if (tile != cur_tile || reads_in_tile + upcoming > kMaxReadsPerSlice) {
      if (cur_tile != 0xFFFFFFFFu) {
        flush(cur_tile);
        __syncthreads();
      }
      zero_lds();
      __syncthreads();
      QK_MARK(1);
      cur_tile = tile;
      reads_in_tile = 0;
    }
    reads_in_tile += process(tile, rb, re);
  };
  if constexpr (ONE) {
    if (blockIdx.x < p.n_slices) run_item(0, (uint64_t)blockIdx.x * p.reads_per_slice, (uint64_t)(blockIdx.x + 1u) * p.reads_per_slice);
  } else
  if (p.static_split) {
    // Several tiles, work known: the read-tiles (tile 0's reads, then tile 1's, ...) are cut into
    // gridDim.x equal shares; a share is a contiguous read range in one tile, sometimes the tail of
    // one and the head of the next — one or two histograms to flush, no queue, no atomics, and as
    // few items as there can be (per-item costs — staging latency, barriers, the flush — were a
    // tenth of the long-read kernel: 1-20 kb reads 0.63 -> 0.55 ms).
    // A tile a share touches costs its workgroup a flush whatever the number of reads (40 quality rows x 512 positions of
    // u64 atomics: ~16 us, the time of ~300 read-tiles), and the last workgroups cross two or three of the thinly
    // populated far tiles: every tile is therefore `tile_overhead` items longer than its reads — items in front of the
    // reads that stand for the flush and are never run.
    const uint64_t F = p.tile_overhead;
    auto prefix = [&](uint32_t t) -> uint64_t { return (p.tile_prefix ? p.tile_prefix[t] : (uint64_t)t * p.n_reads) + (uint64_t)t * F; };
    const uint64_t W = prefix(p.n_tiles);
    uint64_t lo = W / gridDim.x * blockIdx.x + (W % gridDim.x) * blockIdx.x / gridDim.x;
    const uint64_t hi = blockIdx.x + 1u == gridDim.x ? W : W / gridDim.x * (blockIdx.x + 1u) + (W % gridDim.x) * (blockIdx.x + 1u) / gridDim.x;
    // the tile that holds item `lo`: the last t with prefix(t) <= lo
    uint32_t ta = 0, tb = p.n_tiles;   // invariant: prefix(ta) <= lo < prefix(tb) (when lo < W)
    while (tb - ta > 1u) {
      const uint32_t mid = (ta + tb) / 2u;
      if (prefix(mid) <= lo) ta = mid;
      else tb = mid;
    }
    for (uint32_t t = ta; lo < hi && t < p.n_tiles; ++t) {
      const uint64_t t0 = prefix(t) + F, t1 = prefix(t + 1u);   // the tile's reads are items [t0, t1)
      const uint64_t seg_hi = hi < t1 ? hi : t1;
      const uint64_t seg_lo = lo > t0 ? lo : t0;
      if (seg_hi > seg_lo) {
        // (items of at most reads_per_slice reads: the u16 counters of a histogram)
        for (uint64_t r = seg_lo - t0; r < seg_hi - t0; r += p.reads_per_slice) {
          const uint64_t e = r + p.reads_per_slice < seg_hi - t0 ? r + p.reads_per_slice : seg_hi - t0;
          run_item(t, r, e);
        }
      }
      if (seg_hi > lo) lo = seg_hi;
    }
  } else
  if (p.queue == nullptr) {
    // one tile: block b owns read slice b
    if (blockIdx.x < p.n_slices) run_item(0, (uint64_t)blockIdx.x * p.reads_per_slice, (uint64_t)(blockIdx.x + 1u) * p.reads_per_slice);
  } else {
    // several tiles: one queue of read slices per tile.  The workgroups start
    // spread over all tiles (so that a read's tiles are consumed at about the
    // same time by neighbouring workgroups, and flushes hit different table
    // regions) and keep pulling slices of their tile; when it runs dry they
    // move on to the next tile that still has work, flushing only then.
    uint32_t tile = (uint32_t)(((uint64_t)blockIdx.x * p.n_tiles) / gridDim.x);
    if (!FIXED && p.reach) {
      // with the reads sorted by reach the work of a tile is known: home tiles in
      // proportion to it (workgroup b starts where b's share of the read-tiles
      // lies), so that few workgroups ever have to change tile and flush
      if (tid == 0) {
        uint64_t total = 0, acc = 0;
        for (uint32_t t = 0; t < p.n_tiles; ++t) total += p.reach[t];
        const uint64_t want = total * (2ull * blockIdx.x + 1ull) / (2ull * gridDim.x);
        uint32_t t = 0;
        for (; t + 1u < p.n_tiles; ++t) {
          acc += p.reach[t];
          if (acc > want) break;
        }
        lds_misc[1] = t;
      }
      __syncthreads();
      tile = lds_misc[1];
    }
    // Take slices of `tile` while it has any; then move to the next tile
    // (cyclically) that still has work.  The search reads the tiles' counters
    // T at a time (one load per thread) — probing them one by one with an
    // atomic and two barriers each cost a workgroup ~1.5 us per dry tile, i.e.
    // over a millisecond at the end of a launch with ~1000 tiles (3000 reads of
    // 100-500 kb: 1.41 ms).
    for (;;) {
      uint32_t found = 0xFFFFFFFFu;
      for (uint32_t w = 0; w < p.n_tiles && found == 0xFFFFFFFFu; w += T) {
        __syncthreads();   // lds_misc[1] / [3] of the previous round are consumed
        if (tid == 0) lds_misc[3] = 0xFFFFFFFFu;
        __syncthreads();
        const uint32_t off = w + tid;
        bool has = false;
        if (off < p.n_tiles) {
          const uint32_t t = (tile + off) % p.n_tiles;
          const uint32_t taken = __hip_atomic_load(&p.queue[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          has = (uint64_t)taken * p.reads_per_slice < tile_reads(t);
        }
        const uint64_t vote = __builtin_amdgcn_ballot_w64(has);
        if (vote && lane_id == 0) atomicMin(&lds_misc[3], off + (uint32_t)__builtin_ctzll(vote));
        __syncthreads();
        found = lds_misc[3];
      }
      if (found == 0xFFFFFFFFu) break;   // every tile's queue is exhausted
      tile = (tile + found) % p.n_tiles;
      __syncthreads();
      if (tid == 0) lds_misc[1] = atomicAdd(&p.queue[tile], 1u);
      __syncthreads();
      const uint32_t slice = lds_misc[1];
      if ((uint64_t)slice * p.reads_per_slice >= tile_reads(tile)) continue;   // somebody else got the last one
      run_item(tile, (uint64_t)slice * p.reads_per_slice, (uint64_t)(slice + 1u) * p.reads_per_slice);
    }
  }
  if (MODE == 1) {
    if (keep == 0x12345678u) lds[0] = keep;
  }
#if defined(QK_DUMMY_VALU) || defined(QK_ABL)
  if (keep == 0x12345679u) lds[1] = keep;
#endif
  QK_MARK(5);
  if (cur_tile != 0xFFFFFFFFu) flush(cur_tile);
#ifdef QK_TIMING
  __syncthreads();
  QK_MARK(6);
#endif
}

template <int T, int U, bool FIXED, int MODE, bool ADAPT = false, int PD = 1, bool AL = false, bool SV = false, bool W16 = false, bool NP = false, bool ONE = false>
// (strided builds without the adapter scan are held to 64 VGPRs for two workgroups per CU — also the one without tail masks, NP:
// measured with one workgroup 0.5975 ms per 10M trimmed 150 bp reads, with two 0.5178; the masked build 0.5300)
__global__ __launch_bounds__(T, (SV && !ADAPT) ? (T / 256) * 2 : QK_MIN_WAVES_PER_SIMD) void hist_kernel(const HistParams p) {
  hist_body<T, U, FIXED, MODE, ADAPT, PD, AL, SV, W16, NP, ONE>(p);
}

// (A build of the long-read variant held to 120 VGPRs — four of its waves then leave 32 registers of a SIMD free, room for
// the waves of the next batch's reach pre-pass (12-16 VGPRs) on the side stream — was measured, four alternations of 200
// steps on one box: 0.6100 ms per config-5 step against 0.6091 with all 126; not kept.)

// ---- pre-pass of long ragged batches: reads ordered by the tiles they reach --
// bucket k = reads that reach exactly k tiles (k = ceil(len / tile_pos), capped);
// order[] lists the reads by bucket, descending, so the reads that reach tile t
// are order[0 .. reach[t]).  Counting sort: count -> scan -> scatter, both passes
// over the reads privatised in LDS (one global atomic per bucket per block).
constexpr int kReachThreads = 256;
constexpr uint32_t kReachMaxTiles = 4096;
constexpr uint32_t kLenLds = 8192;   // lengths with an LDS counter in the length passes (see ragged_length_kernel)

__device__ __forceinline__ uint32_t tiles_reached(const HistParams &p, uint64_t r) {
  const uint32_t len = p.lengths ? p.lengths[r] : (uint32_t)(p.offsets[r + 1] - p.offsets[r]);
  const uint32_t k = (len + p.tile_pos - 1u) / p.tile_pos;
  return k < p.n_tiles ? k : p.n_tiles;
}

// cnt[key] += 1 for every lane with `valid`, adding lanes that hold the same key up
// first: real batches are dominated by one or two lengths, and 64 lanes adding to
// one LDS address serialise (10M reads, 70 % of one length: 53 us -> 12 us)
__device__ __forceinline__ void wave_count_lds(uint32_t *cnt, uint32_t key, bool valid) {
  uint64_t left = __builtin_amdgcn_ballot_w64(valid);
  const uint32_t lane = threadIdx.x & 63u;
  for (int k = 0; k < 3 && left; ++k) {
    const uint32_t leader = (uint32_t)__builtin_ctzll(left);
    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)leader);
    const uint64_t same = __builtin_amdgcn_ballot_w64(valid && key == v);
    if (lane == leader) atomicAdd(&cnt[v], (uint32_t)__builtin_popcountll(same));
    if (key == v) valid = false;
    left &= ~same;
  }
  if (valid) atomicAdd(&cnt[key], 1u);
}

// the same on a row of the u64 table in global memory (lengths past the LDS counters: long reads
// come in few copies per length — unless the batch is, say, 143k reads of exactly 10,496 bases:
// same-address global atomics serialise at ~15 ns, 1.7 ms for that batch)
__device__ __forceinline__ void wave_count_global(unsigned long long *row, uint32_t key, bool valid) {
  uint64_t left = __builtin_amdgcn_ballot_w64(valid);
  const uint32_t lane = threadIdx.x & 63u;
  for (int k = 0; k < 3 && left; ++k) {
    const uint32_t leader = (uint32_t)__builtin_ctzll(left);
    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)leader);
    const uint64_t same = __builtin_amdgcn_ballot_w64(valid && key == v);
    if (lane == leader) atomicAdd(&row[v], (unsigned long long)__builtin_popcountll(same));
    if (key == v) valid = false;
    left &= ~same;
  }
  if (valid) atomicAdd(&row[key], 1ull);
}

// counts[k] += reads of bucket k (counts: n_tiles + 1 words, zero on entry and
// on exit); the block that finishes last turns the counts into
//   reach[t]  = reads in buckets > t
//   cursor[k] = first slot of bucket k in order[] (longest reads first)
__global__ __launch_bounds__(kReachThreads) void reach_count_kernel(const HistParams p, uint32_t *counts, uint32_t *done,
                                                                    uint32_t *reach, uint32_t *cursor, unsigned long long *prefix) {
  extern __shared__ uint32_t lc[];   // n_tiles + 1 counters, + 1 word for the ticket
  const uint32_t nb = p.n_tiles + 1u;
  for (uint32_t i = threadIdx.x; i < nb; i += kReachThreads) lc[i] = 0;
  __syncthreads();
  // (whole waves per round: wave_count_lds uses ballots.  length_count is hist_kernel's business: it
  // meets every read once in the tile the read ends in)
  for (uint64_t r0 = (uint64_t)blockIdx.x * kReachThreads; r0 < p.n_reads; r0 += (uint64_t)gridDim.x * kReachThreads) {
    const uint64_t r = r0 + threadIdx.x;
    const bool in = r < p.n_reads;
    const uint32_t len = !in ? 0u : (p.lengths ? p.lengths[r] : (uint32_t)(p.offsets[r + 1] - p.offsets[r]));
    const uint32_t k = (len + p.tile_pos - 1u) / p.tile_pos;
    wave_count_lds(lc, k < p.n_tiles ? k : p.n_tiles, in);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < nb; i += kReachThreads)
    if (lc[i]) atomicAdd(&counts[i], lc[i]);
  // (the barrier waits for the atomics; the ticket's release orders them for the last block.  A __threadfence() by every
  // thread here was most of this kernel's 15 us: an agent-scope fence writes back and invalidates the XCD's L2)
  __syncthreads();
  if (threadIdx.x == 0) lc[nb] = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (lc[nb] != gridDim.x - 1u) return;
  // last block: every other block's counts are in (their atomics precede their ticket)
  for (uint32_t i = threadIdx.x; i < nb; i += kReachThreads) lc[i] = atomicExch(&counts[i], 0u);
  if (threadIdx.x == 0) *done = 0;
  __syncthreads();
  // lc[k] <- reads in buckets > k (k = 0 .. n_tiles): a suffix sum, then prefix[t] = sum of lc[0 .. t) — both as
  // block-wide scans (thread 0 alone walked the buckets twice: 40 tiles cost 5 us of a 15 us kernel, 1000 tiles 65)
  __shared__ unsigned long long part[kReachThreads];
  const uint32_t per = (nb + kReachThreads - 1u) / kReachThreads;
  const uint32_t lo = threadIdx.x * per < nb ? threadIdx.x * per : nb, hi = lo + per < nb ? lo + per : nb;
  auto scan_partials = [&](bool from_the_end) {   // part[i] <- sum of the partials from i to the end (from the start to i), inclusive
    __syncthreads();
    for (uint32_t d = 1; d < kReachThreads; d <<= 1) {
      const uint32_t j = from_the_end ? threadIdx.x + d : threadIdx.x - d;
      const unsigned long long add = (from_the_end ? j < kReachThreads : threadIdx.x >= d) ? part[j] : 0ull;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
  };
  {
    unsigned long long sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += lc[k];
    part[threadIdx.x] = sum;
    scan_partials(true);                      // inclusive: own segment and everything behind it
    unsigned long long above = part[threadIdx.x] - sum;   // buckets behind this thread's segment
    for (uint32_t k = hi; k > lo; --k) {
      const uint32_t c = lc[k - 1u];
      lc[k - 1u] = (uint32_t)above;           // reads in buckets > k - 1
      above += c;
    }
    __syncthreads();
  }
  for (uint32_t t = threadIdx.x; t < p.n_tiles; t += kReachThreads) {
    reach[t] = lc[t];                        // reads in buckets > t
    cursor[t + 1u] = lc[t + 1u];             // bucket t + 1 starts behind all longer reads
  }
  {                                          // prefix[t] = read-tiles of the tiles before t (hist_kernel's static split)
    const uint32_t nt = p.n_tiles;
    const uint32_t l2 = lo < nt ? lo : nt, h2 = hi < nt ? hi : nt;
    unsigned long long sum = 0;
    for (uint32_t t = l2; t < h2; ++t) sum += lc[t];
    part[threadIdx.x] = sum;
    scan_partials(false);
    unsigned long long acc = part[threadIdx.x] - sum;
    for (uint32_t t = l2; t < h2; ++t) {
      prefix[t] = acc;
      acc += lc[t];
    }
    if (threadIdx.x == kReachThreads - 1u) prefix[nt] = acc;   // (the last thread's segment ends the tiles, or is empty behind them)
  }
}

// order[]: block b owns a contiguous share of the reads; it counts its share per
// bucket, reserves one range per bucket (one global atomic each), then places
// its reads with LDS cursors.  (Same-address global atomics serialise at ~15 ns:
// per-read or per-round reservations cost more than the whole rest of the pass.)
__global__ __launch_bounds__(kReachThreads) void reach_scatter_kernel(const HistParams p, uint32_t *cursor, uint32_t *order) {
  extern __shared__ uint32_t lc[];
  const uint32_t nb = p.n_tiles + 1u;
  const uint64_t share = (p.n_reads + gridDim.x - 1) / gridDim.x;
  const uint64_t r0 = (uint64_t)blockIdx.x * share, r1 = r0 + share < p.n_reads ? r0 + share : p.n_reads;
  for (uint32_t i = threadIdx.x; i < nb; i += kReachThreads) lc[i] = 0;
  __syncthreads();
  for (uint64_t r = r0 + threadIdx.x; r < r1; r += kReachThreads) atomicAdd(&lc[tiles_reached(p, r)], 1u);
  __syncthreads();
  for (uint32_t i = 1u + threadIdx.x; i < nb; i += kReachThreads)
    if (lc[i]) lc[i] = atomicAdd(&cursor[i], lc[i]);   // count -> first slot of the block's range
  __syncthreads();
  for (uint64_t r = r0 + threadIdx.x; r < r1; r += kReachThreads) {
    const uint32_t k = tiles_reached(p, r);
    if (k) order[atomicAdd(&lc[k], 1u)] = (uint32_t)r;
  }
}

// length_count and the kmers==NULL count (quack.c:215-219) of a ragged batch
// that spans several tiles.  Inside hist_kernel only lengths below the tile
// width have an LDS counter; longer ones would be one global atomic per read —
// 5M reads of ~600 bases on 20 addresses: 11.3 ms instead of 1.3.  Here every
// block privatises the first 8192 lengths in LDS (longer reads come in few
// copies per length) and flushes what is non-zero.
// (1024 threads per block and one block per CU: every block ends with one global atomic per
// length it met, and same-address atomics serialise at ~15 ns — 2048 blocks cost 30 us there)
constexpr int kLenThreads = 1024;
__global__ __launch_bounds__(kLenThreads) void ragged_length_kernel(const HistParams p) {
  __shared__ uint32_t cnt[kLenLds];
  __shared__ uint32_t gt10;
  for (uint32_t i = threadIdx.x; i < kLenLds; i += kLenThreads) cnt[i] = 0;
  if (threadIdx.x == 0) gt10 = 0;
  __syncthreads();
  uint32_t mine = 0;
  auto one = [&](uint32_t len, bool in) {
    mine += (in && len > 10u) ? 1u : 0u;
    const uint32_t lp = len - 1u;
    wave_count_lds(cnt, lp, in && len != 0 && lp < kLenLds);
    // (a length beyond the table can only come from device-side lengths[] the host never saw: it is
    // reported through the status word — the next sync fails — and never written)
    if (in && len != 0 && (lp >= p.table_len || (p.len_limit && len > p.len_limit))) atomicOr(p.status, kStatusBadLength);
    wave_count_global(&p.table[(uint64_t)kRowLength * p.table_len], lp, in && len != 0 && lp >= kLenLds && lp < p.table_len);
  };
  // (whole waves per round: wave_count_lds uses ballots)
  if (p.lengths && (reinterpret_cast<uintptr_t>(p.lengths) & 15u) == 0) {
    // four lengths per lane and load: one dword per lane in flight moves 40 MB of lengths
    // (10M reads) at a twentieth of the memory rate (measured 85 us)
    const uint4 *l4 = reinterpret_cast<const uint4 *>(p.lengths);
    const uint64_t groups = (p.n_reads + 3) / 4;
    for (uint64_t g0 = (uint64_t)blockIdx.x * kLenThreads; g0 < groups; g0 += (uint64_t)gridDim.x * kLenThreads) {
      const uint64_t g = g0 + threadIdx.x;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g * 4 + 3 < p.n_reads) {
        v = l4[g];
      } else if (g < groups) {
        const uint32_t *tail = p.lengths + g * 4;
        const uint64_t left = p.n_reads - g * 4;
        v.x = tail[0];
        if (left > 1) v.y = tail[1];
        if (left > 2) v.z = tail[2];
      }
      one(v.x, g * 4 < p.n_reads);
      one(v.y, g * 4 + 1 < p.n_reads);
      one(v.z, g * 4 + 2 < p.n_reads);
      one(v.w, g * 4 + 3 < p.n_reads);
    }
  } else {
    for (uint64_t r0 = (uint64_t)blockIdx.x * kLenThreads; r0 < p.n_reads; r0 += (uint64_t)gridDim.x * kLenThreads) {
      const uint64_t r = r0 + threadIdx.x;
      const bool in = r < p.n_reads;
      one(!in ? 0u : (p.lengths ? p.lengths[r] : (uint32_t)(p.offsets[r + 1] - p.offsets[r])), in);
    }
  }
  if (mine) atomicAdd(&gt10, mine);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kLenLds && i < p.table_len; i += kLenThreads)
    if (cnt[i]) atomicAdd(&p.table[(uint64_t)kRowLength * p.table_len + i], (unsigned long long)cnt[i]);
  if (threadIdx.x == 0 && p.no_adapters && gt10)
    atomicAdd(&p.table[(uint64_t)kRowKmer * p.table_len + 10u], (unsigned long long)gt10);
}

// The same for batches whose reads are at most kShortLen long (strided batches of
// trimmed short reads: millions of reads on a few dozen lengths, most of them on
// ONE).  Counters [length][32]: a lane adds into column lane % 32, so the LDS
// atomics of a wave never share a bank, let alone an address (two lanes at
// most) — no ballots, no serialisation: 10M reads 48 us -> ~8 us.
constexpr uint32_t kShortLen = 512;
__global__ __launch_bounds__(kLenThreads) void short_length_kernel(const HistParams p) {
  __shared__ uint32_t cnt[kShortLen * 32u];
  __shared__ uint32_t gt10;
  for (uint32_t i = threadIdx.x; i < kShortLen * 32u; i += kLenThreads) cnt[i] = 0;
  if (threadIdx.x == 0) gt10 = 0;
  __syncthreads();
  const uint32_t col = threadIdx.x & 31u;
  uint32_t mine = 0;
  auto one = [&](uint32_t len) {
    mine += len > 10u ? 1u : 0u;
    // strided batches: no read is longer than the caller declared (<= the stride; the table holds that many positions)
    if (len > p.len_limit || len > p.table_len) atomicOr(p.status, kStatusBadLength);
    else if (len != 0 && len <= kShortLen) atomicAdd(&cnt[(len - 1u) * 32u + col], 1u);
    else if (len != 0) atomicAdd(&p.table[(uint64_t)kRowLength * p.table_len + len - 1u], 1ull);   // (not expected)
  };
  if ((reinterpret_cast<uintptr_t>(p.lengths) & 15u) == 0) {
    const uint4 *l4 = reinterpret_cast<const uint4 *>(p.lengths);
    const uint64_t groups = p.n_reads / 4;
    for (uint64_t g = (uint64_t)blockIdx.x * kLenThreads + threadIdx.x; g < groups; g += (uint64_t)gridDim.x * kLenThreads) {
      const uint4 v = l4[g];
      one(v.x);
      one(v.y);
      one(v.z);
      one(v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (p.n_reads & 3u)) one(p.lengths[groups * 4 + threadIdx.x]);
  } else {
    for (uint64_t r = (uint64_t)blockIdx.x * kLenThreads + threadIdx.x; r < p.n_reads; r += (uint64_t)gridDim.x * kLenThreads)
      one(p.lengths[r]);
  }
  if (mine) atomicAdd(&gt10, mine);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kShortLen && i < p.table_len; i += kLenThreads) {
    uint32_t c = 0;
    for (uint32_t k = 0; k < 32u; ++k) c += cnt[i * 32u + ((k + i) & 31u)];   // (staggered: no bank conflicts)
    if (c) atomicAdd(&p.table[(uint64_t)kRowLength * p.table_len + i], (unsigned long long)c);
  }
  if (threadIdx.x == 0 && p.no_adapters && gt10)
    atomicAdd(&p.table[(uint64_t)kRowKmer * p.table_len + 10u], (unsigned long long)gt10);
}

// starts[i] = (i % per_chunk) * stride: a strided batch restated as gapped ones (launch geometries
// the strided kernel variant is not built for; every chunk of per_chunk reads is addressed from its own base)
// (and lengths[] is vetted on the way, as the strided length kernels do: the gapped kernels trust their lengths)
__global__ __launch_bounds__(256) void strided_starts_kernel(unsigned long long *starts, uint64_t n, uint64_t per_chunk, uint32_t stride,
                                                             const uint32_t *lengths, uint32_t len_limit, uint32_t *status) {
  bool bad = false;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    starts[i] = (i % per_chunk) * stride;
    bad |= lengths[i] > len_limit;
  }
  if (bad) atomicOr(status, kStatusBadLength);
}

// QK_BATCH_NEUTRAL_PADS on a device-resident batch is a promise the kernels take at its word; QUACK_HIP_CHECK_PADS=1 makes the
// shim look first (tests, debugging a producer): every byte of [length, stride) of every read, both arrays
__global__ __launch_bounds__(256) void pads_check_kernel(const uint8_t *seq, const uint8_t *qual, const uint32_t *lengths, uint64_t n,
                                                         uint32_t stride, uint32_t *status) {
  bool bad = false;
  for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (uint64_t)gridDim.x * 256) {
    const uint32_t len = lengths[r] < stride ? lengths[r] : stride;
    for (uint32_t k = len; k < stride; ++k) bad |= seq[r * stride + k] != 0xFFu || qual[r * stride + k] != 0xFFu;
  }
  if (bad) atomicOr(status, kStatusBadPads);
}

// table += table32, table32 = 0 (HistParams::table32)
__global__ __launch_bounds__(256) void table_fold_kernel(unsigned long long *table, uint32_t *table32, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint32_t v = table32[i];
    if (v) {
      table[i] += v;
      table32[i] = 0;
    }
  }
}

// dst += src over the planar tables of two accumulators on the same device
__global__ __launch_bounds__(256) void table_add_kernel(unsigned long long *dst, const unsigned long long *src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] += src[i];
}

}  // namespace qk
