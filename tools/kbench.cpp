// kbench — kernel exploration harness (developer tool, not part of the
// product path and not used by bench.py).  Generates a device-resident
// synthetic batch, sweeps launch configurations / ablation modes of the
// histogram kernel through the C-ABI, and checks MODE 0 against a direct host
// count of the same bytes.
//
//   kbench [n_reads] [read_len] [ragged:0|1] [adapters:0|1]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "quack_hip.h"

extern "C" int qk_debug_set_mode(int mode);

#define CK(x)                                                                  \
  do {                                                                         \
    int rc_ = (x);                                                             \
    if (rc_) {                                                                 \
      fprintf(stderr, "FAIL %s -> %d: %s\n", #x, rc_, qk_last_error());        \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)
#define HK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "HIP FAIL %s: %s\n", #x, hipGetErrorString(e_));         \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)

// reference point: plain streaming read of both arrays, 16 B/lane, grid-stride
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4 *a, const uint4 *b, size_t n16, unsigned *sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (; i + 256 * (UNROLL - 1) < n16; i += stride) {
    uint4 x[UNROLL], y[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { x[u] = a[i + 256 * u]; y[u] = b[i + 256 * u]; }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      acc.x ^= x[u].x ^ y[u].x; acc.y ^= x[u].y ^ y[u].y; acc.z ^= x[u].z ^ y[u].z; acc.w ^= x[u].w ^ y[u].w;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

// variants isolating what the hist kernel's load pattern costs:
//  W bytes per lane (8 or 16), T threads, each workgroup streams a contiguous
//  slice (like hist_kernel) or the grid strides; `skew` shifts every row of
//  `row` bytes like unaligned 150-byte reads do.
template <int W, int T, int U, bool SLICE>
__global__ __launch_bounds__(T) void stream_var_kernel(const uint8_t *a, const uint8_t *b, size_t bytes, unsigned *sink) {
  unsigned acc = 0;
  const size_t per_iter = (size_t)T * W * U;
  size_t n_iter = bytes / per_iter;
  size_t it0, it1, step;
  if (SLICE) {
    size_t per_wg = (n_iter + gridDim.x - 1) / gridDim.x;
    it0 = blockIdx.x * per_wg; it1 = it0 + per_wg < n_iter ? it0 + per_wg : n_iter; step = 1;
  } else { it0 = blockIdx.x; it1 = n_iter; step = gridDim.x; }
  for (size_t it = it0; it < it1; it += step) {
    const size_t base = it * per_iter + (size_t)threadIdx.x * W;
    if (W == 16) {
      uint4 x[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { __builtin_memcpy(&x[u], a + base + (size_t)u * T * W, 16); __builtin_memcpy(&y[u], b + base + (size_t)u * T * W, 16); }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w ^ y[u].x ^ y[u].y ^ y[u].z ^ y[u].w;
    } else {
      uint2 x[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { __builtin_memcpy(&x[u], a + base + (size_t)u * T * W, 8); __builtin_memcpy(&y[u], b + base + (size_t)u * T * W, 8); }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ y[u].x ^ y[u].y;
    }
  }
  if (acc == 0x12345678u) *sink = 1;
}

// each lane loads a 16-byte window at an 8-byte stride (windows overlap by 8 B:
// every word is fetched by two lanes; HBM traffic unchanged, L1 traffic x2)
template <int T, int U, int LW>
__global__ __launch_bounds__(T) void stream_overlap_kernel(const uint8_t *a, const uint8_t *b, size_t bytes, unsigned *sink) {
  unsigned acc = 0;
  const size_t per_iter = (size_t)T * 8 * U;
  const size_t n_iter = bytes / per_iter;
  const size_t per_wg = (n_iter + gridDim.x - 1) / gridDim.x;
  const size_t it0 = blockIdx.x * per_wg, it1 = it0 + per_wg < n_iter ? it0 + per_wg : n_iter;
  for (size_t it = it0; it < it1; ++it) {
    const size_t base = it * per_iter + (size_t)threadIdx.x * 8;
    if (LW == 16) {
      uint4 x[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { __builtin_memcpy(&x[u], a + base + (size_t)u * T * 8, 16); __builtin_memcpy(&y[u], b + base + (size_t)u * T * 8, 16); }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w ^ y[u].x ^ y[u].y ^ y[u].z ^ y[u].w;
    } else {  // 12-byte window: dwordx3
      struct u3 { unsigned x, y, z; };
      u3 x[U], y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { __builtin_memcpy(&x[u], a + base + (size_t)u * T * 8, 12); __builtin_memcpy(&y[u], b + base + (size_t)u * T * 8, 12); }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ y[u].x ^ y[u].y ^ y[u].z;
    }
  }
  if (acc == 0x12345678u) *sink = 1;
}

template <int T, int U, int LW>
static void run_overlap(const char *name, const uint8_t *a, const uint8_t *b, size_t bytes, unsigned *sink, int grid) {
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    HK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream_overlap_kernel<T, U, LW>), dim3(grid), dim3(T), 0, 0, a, b, bytes, sink);
    HK(hipEventRecord(e1, 0));
    HK(hipEventSynchronize(e1));
    float ms; HK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("  %-44s grid %5d: %.3f ms  %.2f TB/s\n", name, grid, ms / 20, 2.0 * bytes / (ms / 20 * 1e-3) / 1e12);
  }
}

template <int W, int T, int U, bool SLICE>
static void run_var(const char *name, const uint8_t *a, const uint8_t *b, size_t bytes, unsigned *sink, int grid) {
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    HK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream_var_kernel<W, T, U, SLICE>), dim3(grid), dim3(T), 0, 0, a, b, bytes, sink);
    HK(hipEventRecord(e1, 0));
    HK(hipEventSynchronize(e1));
    float ms; HK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("  %-44s grid %5d: %.3f ms  %.2f TB/s\n", name, grid, ms / 20, 2.0 * bytes / (ms / 20 * 1e-3) / 1e12);
  }
}

static int base_code(uint8_t c) {
  unsigned k = c & 31u;
  return k == 20 ? 1 : k == 3 ? 2 : k == 7 ? 3 : 0;
}

static inline uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}


int main(int argc, char **argv) {
  uint64_t n_reads = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
  uint32_t read_len = argc > 2 ? (uint32_t)atoi(argv[2]) : 150;
  int ragged = argc > 3 ? atoi(argv[3]) : 0;
  int adapters = argc > 4 ? atoi(argv[4]) : 0;
  int only_cfg = argc > 5 ? atoi(argv[5]) : -1;   // -1: sweep all
  int only_mode = argc > 6 ? atoi(argv[6]) : -1;
  int skip_check = argc > 7 ? atoi(argv[7]) : 0;
  int h2d = argc > 8 ? atoi(argv[8]) : 0;   // 1: also time the pinned H2D pipeline

  // ---- synthetic batch -------------------------------------------------
  // KB_ALIGNED=1 (ragged): every read starts on a 128-byte line (gapped batch, QK_BATCH_ALIGNED128) —
  // the layout the host feed gives long reads
  const bool gapped = ragged && getenv("KB_ALIGNED") != nullptr;
  std::vector<uint64_t> off(n_reads + 1);
  std::vector<uint32_t> lens(n_reads);
  uint64_t seed = 2, total = 0, n_bases = 0;
  uint32_t max_len = 0;
  for (uint64_t r = 0; r < n_reads; ++r) {
    if (gapped) total = (total + 127) & ~127ull;
    off[r] = total;
    uint32_t l = read_len;
    if (ragged) {
      const uint32_t lo = getenv("KB_RAGGED_LO") ? (uint32_t)atoi(getenv("KB_RAGGED_LO")) : 1u;   // lengths uniform in [lo, read_len]
      l = lo + (uint32_t)(splitmix(seed) % (read_len - lo + 1));
    }
    if (l > max_len) max_len = l;
    lens[r] = l;
    total += l;
    n_bases += l;
  }
  off[n_reads] = total;
  std::vector<uint8_t> seq(total + QK_TAIL_SLACK, 0), qual(total + QK_TAIL_SLACK, 0);
  static const char B[4] = {'A', 'C', 'G', 'T'};
  for (uint64_t i = 0; i < total; i += 8) {
    uint64_t a = splitmix(seed), b = splitmix(seed);
    for (int k = 0; k < 8 && i + k < total; ++k) {
      seq[i + k] = B[(a >> (8 * k)) & 3];
      qual[i + k] = (uint8_t)(33 + 2 + ((((b >> (8 * k)) & 0xFF) * 40) >> 8));
    }
  }
  std::vector<uint32_t> bits;
  if (adapters) {
    bits.assign(QK_KMER_TABLE_WORDS, 0);
    const double splice = getenv("KB_SPLICE") ? atof(getenv("KB_SPLICE")) : 0.0;
    if (splice > 0 && !ragged) {
      // config 3's shape: 24 adapters of 30-60 nt (read_adapters rule: the first 10-mer of a record is not
      // inserted, quack.c:164-172), a fraction of the reads gets one at a uniform offset, cut at the read end
      std::vector<std::vector<uint8_t>> ads(24);
      for (auto &a : ads) {
        a.resize(30 + splitmix(seed) % 31);
        for (auto &c : a) c = (uint8_t)B[splitmix(seed) & 3];
        uint32_t idx = 0;
        for (size_t i = 0; i < a.size(); ++i) {
          idx = ((idx << 2) + (uint32_t)base_code(a[i])) & 0xFFFFF;
          if (i >= 10) bits[idx >> 5] |= 1u << (idx & 31);
        }
      }
      uint64_t n_spliced = 0;
      for (uint64_t r = 0; r < n_reads; ++r) {
        if ((double)(splitmix(seed) >> 11) * (1.0 / 9007199254740992.0) >= splice) continue;
        const auto &a = ads[splitmix(seed) % ads.size()];
        const uint32_t at = (uint32_t)(splitmix(seed) % read_len);
        for (uint32_t i = 0; i < a.size() && at + i < read_len; ++i) seq[off[r] + at + i] = a[i];
        ++n_spliced;
      }
      printf("spliced an adapter into %llu reads\n", (unsigned long long)n_spliced);
    } else {
      for (int i = 0; i < 333; ++i) {
        uint32_t km = (uint32_t)(splitmix(seed) & 0xFFFFF);
        bits[km >> 5] |= 1u << (km & 31);
      }
    }
  }
  printf("batch: %llu reads, %llu bases, max_len %u, ragged=%d adapters=%d%s\n",
         (unsigned long long)n_reads, (unsigned long long)n_bases, max_len, ragged, adapters, gapped ? " (reads on 128-byte lines)" : "");

  uint8_t *d_seq, *d_qual;
  uint64_t *d_off = nullptr;
  HK(hipMalloc((void **)&d_seq, total + QK_TAIL_SLACK));
  size_t qual_shift = getenv("KB_QUAL_SHIFT") ? strtoull(getenv("KB_QUAL_SHIFT"), 0, 10) : 0;
  HK(hipMalloc((void **)&d_qual, total + QK_TAIL_SLACK + qual_shift));
  d_qual += qual_shift;
  printf("d_seq %p d_qual %p (shift %zu)\n", (void *)d_seq, (void *)d_qual, qual_shift);
  HK(hipMemcpy(d_seq, seq.data(), total + QK_TAIL_SLACK, hipMemcpyHostToDevice));
  HK(hipMemcpy(d_qual, qual.data(), total + QK_TAIL_SLACK, hipMemcpyHostToDevice));
  uint32_t *d_len = nullptr;
  if (ragged) {
    HK(hipMalloc((void **)&d_off, (n_reads + 1) * 8));
    HK(hipMemcpy(d_off, off.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice));
    HK(hipMalloc((void **)&d_len, n_reads * 4));
    HK(hipMemcpy(d_len, lens.data(), n_reads * 4, hipMemcpyHostToDevice));
  }
  auto submit = [&](qk_accum *acc) {
    return gapped ? qk_accum_submit_device_gapped(acc, d_seq, d_qual, d_off, d_len, n_reads, total, max_len, QK_BATCH_ALIGNED128, nullptr)
                  : qk_accum_submit_device(acc, d_seq, d_qual, d_off, n_reads, total, max_len, nullptr);
  };

  // ---- correctness: one submit on a fresh accumulator vs a host count ----
  if (!skip_check) {
    qk_accum *acc;
    CK(qk_accum_create(&acc, 0, adapters ? bits.data() : nullptr, max_len));
    qk_debug_set_mode(0);
    CK(submit(acc));
    std::vector<qk_base_info> got(max_len);
    uint64_t ml, nr;
    CK(qk_accum_finish(acc, got.data(), max_len, &ml, &nr));
    std::vector<qk_base_info> want(max_len);
    memset(want.data(), 0, max_len * sizeof(qk_base_info));
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t r = 0; r < n_reads; ++r) {
      const uint8_t *s = &seq[off[r]], *q = &qual[off[r]];
      uint32_t l = lens[r];
      for (uint32_t i = 0; i < l; ++i) {
        want[i].content[base_code(s[i])]++;
        unsigned b = q[i] & 127u;
        if (b >= 33 && b <= 123) want[i].scores[b - 33]++;
      }
      uint32_t i = 10;
      if (adapters) {
        uint32_t idx = 0;
        for (uint32_t k = 0; k < 10 && k < l; ++k) idx = ((idx << 2) + base_code(s[k])) & 0xFFFFF;
        for (; l >= 10 && !((bits[idx >> 5] >> (idx & 31)) & 1) && i < l; ++i)
          idx = ((idx << 2) + base_code(s[i])) & 0xFFFFF;
      }
      if (i < l) want[i].kmer_count++;
      if (l) want[l - 1].length_count++;
    }
    double cpu_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t bad = 0;
    const uint64_t *g = (const uint64_t *)got.data(), *w = (const uint64_t *)want.data();
    for (uint64_t i = 0; i < (uint64_t)max_len * QK_N_ROWS; ++i)
      if (g[i] != w[i]) {
        if (bad < 10)
          printf("  MISMATCH pos %llu row %llu: got %llu want %llu\n", (unsigned long long)(i / QK_N_ROWS),
                 (unsigned long long)(i % QK_N_ROWS), (unsigned long long)g[i], (unsigned long long)w[i]);
        ++bad;
      }
    printf("check: max_len %llu n_reads %llu mismatches %llu  (host count %.2f s = %.1f Mbases/s)\n",
           (unsigned long long)ml, (unsigned long long)nr, (unsigned long long)bad, cpu_s, total / cpu_s / 1e6);
    qk_accum_destroy(acc);
    if (bad) return 1;
  }

  // ---- streaming-read reference point ------------------------------------
  if (!ragged) {
    unsigned *d_sink;
    HK(hipMalloc((void **)&d_sink, 4));
    const size_t n16 = total / 16;
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0));
    HK(hipEventCreate(&e1));
    for (int grid : {2048, 4096, 8192}) {
      for (int rep = 0; rep < 2; ++rep) {
        HK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i)
          hipLaunchKernelGGL(stream_read_kernel<4>, dim3(grid), dim3(256), 0, 0, (const uint4 *)d_seq, (const uint4 *)d_qual, n16, d_sink);
        HK(hipEventRecord(e1, 0));
        HK(hipEventSynchronize(e1));
        float ms;
        HK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("stream_read 16B/lane grid=%d: %.3f ms per pass, %.2f TB/s\n", grid, ms / 10, 2.0 * n16 * 16 / (ms / 10 * 1e-3) / 1e12);
      }
    }
  }

  if (!ragged && getenv("KB_STREAM_VARIANTS")) {
    unsigned *d_sink; HK(hipMalloc((void **)&d_sink, 4));
    size_t bytes = total / 65536 * 65536;
    run_var<16, 256, 4, false>("16B/lane T=256 U=4 grid-stride", d_seq, d_qual, bytes, d_sink, 2048);
    run_var<8, 256, 4, false>("8B/lane T=256 U=4 grid-stride", d_seq, d_qual, bytes, d_sink, 2048);
    run_var<8, 256, 8, false>("8B/lane T=256 U=8 grid-stride", d_seq, d_qual, bytes, d_sink, 2048);
    run_var<8, 1024, 4, false>("8B/lane T=1024 U=4 grid-stride", d_seq, d_qual, bytes, d_sink, 512);
    run_var<8, 1024, 4, true>("8B/lane T=1024 U=4 slice/WG", d_seq, d_qual, bytes, d_sink, 512);
    run_var<16, 1024, 4, true>("16B/lane T=1024 U=4 slice/WG", d_seq, d_qual, bytes, d_sink, 512);
    run_var<16, 1024, 2, true>("16B/lane T=1024 U=2 slice/WG", d_seq, d_qual, bytes, d_sink, 512);
    run_var<8, 1024, 4, true>("8B/lane T=1024 U=4 slice/WG +6B skew", d_seq + 6, d_qual + 6, bytes - 65536, d_sink, 512);
    run_var<16, 1024, 4, true>("16B/lane T=1024 U=4 slice/WG +6B skew", d_seq + 6, d_qual + 6, bytes - 65536, d_sink, 512);
    run_var<8, 1024, 8, true>("8B/lane T=1024 U=8 slice/WG", d_seq, d_qual, bytes, d_sink, 512);
    run_var<8, 512, 4, true>("8B/lane T=512 U=4 slice/WG", d_seq, d_qual, bytes, d_sink, 1024);
    run_var<8, 1024, 4, true>("8B/lane T=1024 U=4 slice/WG grid 256", d_seq, d_qual, bytes, d_sink, 256);
    run_overlap<1024, 4, 16>("16B window @8B stride (aligned 8) U=4", d_seq, d_qual, bytes - 65536, d_sink, 512);
    run_overlap<1024, 2, 16>("16B window @8B stride (aligned 8) U=2", d_seq, d_qual, bytes - 65536, d_sink, 512);
    run_overlap<1024, 4, 12>("12B window @8B stride (aligned 8) U=4", d_seq, d_qual, bytes - 65536, d_sink, 512);
    run_overlap<1024, 4, 12>("12B window @8B stride (aligned 4: +4)", d_seq + 4, d_qual + 4, bytes - 65536, d_sink, 512);
    run_overlap<512, 4, 16>("16B window @8B stride T=512 U=4", d_seq, d_qual, bytes - 65536, d_sink, 1024);
  }

  // ---- sweep -------------------------------------------------------------
  struct Cfg { int T, U, tile, wgs; };
  std::vector<Cfg> cfgs = {
      {1024, 4, 192, 2}, {1024, 2, 192, 2}, {1024, 1, 192, 2}, {512, 4, 192, 4}, {512, 2, 192, 4},
      {256, 4, 192, 8},  {1024, 4, 192, 1}, {1024, 4, 192, 4}, {512, 4, 192, 8}, {1024, 4, 304, 1},
      {512, 4, 304, 2},  {1024, 4, 96, 2},  {1024, 0, 0, 0} /* 12: planner defaults */,
      {1024, 2, 0, 0}, {1024, 1, 0, 0}, {512, 4, 0, 0}, {512, 2, 0, 0} /* 13-16: planner tiles, other T/U */,
      {512, 1, 0, 0}, {512, 1, 0, 3}, {512, 2, 0, 3}, {256, 1, 0, 6}, {256, 2, 0, 5} /* 17-21 */,
      {1024, 1, 0, 2}, {512, 1, 0, 4}, {512, 1, 0, 2} /* 22-24 */,
  };
  const int modes_fixed[] = {0, 1, 2, 3};
  const double alg_bytes = 2.0 * n_bases + (ragged ? (gapped ? 12.0 : 8.0) * n_reads : 0.0);
  for (size_t ci = 0; ci < cfgs.size(); ++ci) {
    const Cfg &c = cfgs[ci];
    if (only_cfg >= 0 && (int)ci != only_cfg) continue;
    for (int mode : modes_fixed) {
      if ((ragged || adapters) && mode != 0) continue;
      if (only_mode >= 0 && mode != only_mode) continue;
      qk_accum *acc;
      CK(qk_accum_create(&acc, 0, adapters ? bits.data() : nullptr, max_len));
      CK(qk_accum_configure(acc, c.T, c.U, c.tile, c.wgs));
      qk_debug_set_mode(mode);
      int rc = submit(acc);
      if (rc) {
        printf("T=%4d U=%d tile=%3d wgs=%d mode=%d : skipped (%s)\n", c.T, c.U, c.tile, c.wgs, mode, qk_last_error());
        qk_accum_destroy(acc);
        continue;
      }
      // warm up for ~100 ms so the clocks have ramped before anything is timed
      for (int i = 0; i < (int)(0.1 / (total * 4e-13)) + 2; ++i)
        CK(submit(acc));
      CK(qk_accum_sync(acc));
      CK(qk_accum_timing_enable(acc, 1));
      const int iters = 20;
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters; ++i)
        CK(submit(acc));
      CK(qk_accum_sync(acc));
      double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters;
      double ms;
      uint64_t launches;
      CK(qk_accum_timing_read(acc, &ms, &launches));
      double k = ms / launches * 1e-3;
      printf("T=%4d U=%d tile=%3d wgs=%d mode=%d : hist kernel %.3f ms  %.2f TB/s (%.1f%% of 8)  %.1f Gbases/s | wall/step %.3f ms\n",
             c.T, c.U, c.tile, c.wgs, mode, k * 1e3, alg_bytes / k / 1e12, alg_bytes / k / 8e12 * 100,
             n_bases / k / 1e9, wall * 1e3);
      fflush(stdout);
      qk_accum_destroy(acc);
    }
  }
  qk_debug_set_mode(0);

  // ---- H2D-inclusive: pinned double-buffered slots, hipMemcpyAsync + kernels
  if (h2d && !ragged) {
    qk_accum *acc;
    CK(qk_accum_create(&acc, 0, adapters ? bits.data() : nullptr, max_len));
    uint8_t *hs, *hq;
    uint64_t *ho, capb, capr;
    const int rounds = 40;
    uint64_t per = 0;
    for (int i = 0; i < rounds + 2; ++i) {
      CK(qk_accum_acquire(acc, &hs, &hq, &ho, &capb, &capr));
      per = capb / read_len;
      if (i < 2) {  // fill each slot once with real reads (afterwards the bytes stay)
        memcpy(hs, seq.data(), per * read_len);
        memcpy(hq, qual.data(), per * read_len);
      }
      CK(qk_accum_commit(acc, per, per * read_len, 0, read_len));
      if (i == 1) CK(qk_accum_sync(acc));
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < rounds; ++i) {
      CK(qk_accum_acquire(acc, &hs, &hq, &ho, &capb, &capr));
      CK(qk_accum_commit(acc, per, per * read_len, 0, read_len));
    }
    CK(qk_accum_sync(acc));
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double bases = (double)rounds * per * read_len;
    printf("h2d pipeline: %d batches x %llu reads (%.0f MiB/array): %.2f Gbases/s, %.1f GB/s over PCIe\n", rounds,
           (unsigned long long)per, per * read_len / 1048576.0, bases / dt / 1e9, 2 * bases / dt / 1e9);
    qk_accum_destroy(acc);
  }
  return 0;
}
