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

static inline uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static int base_code(uint8_t c) {
  unsigned k = c & 31u;
  return k == 20 ? 1 : k == 3 ? 2 : k == 7 ? 3 : 0;
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
  std::vector<uint64_t> off(n_reads + 1);
  uint64_t seed = 2, total = 0;
  uint32_t max_len = 0;
  for (uint64_t r = 0; r < n_reads; ++r) {
    off[r] = total;
    uint32_t l = read_len;
    if (ragged) l = 1 + (uint32_t)(splitmix(seed) % read_len);
    if (l > max_len) max_len = l;
    total += l;
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
    for (int i = 0; i < 333; ++i) {
      uint32_t km = (uint32_t)(splitmix(seed) & 0xFFFFF);
      bits[km >> 5] |= 1u << (km & 31);
    }
  }
  printf("batch: %llu reads, %llu bases, max_len %u, ragged=%d adapters=%d\n",
         (unsigned long long)n_reads, (unsigned long long)total, max_len, ragged, adapters);

  uint8_t *d_seq, *d_qual;
  uint64_t *d_off = nullptr;
  HK(hipMalloc((void **)&d_seq, total + QK_TAIL_SLACK));
  HK(hipMalloc((void **)&d_qual, total + QK_TAIL_SLACK));
  HK(hipMemcpy(d_seq, seq.data(), total + QK_TAIL_SLACK, hipMemcpyHostToDevice));
  HK(hipMemcpy(d_qual, qual.data(), total + QK_TAIL_SLACK, hipMemcpyHostToDevice));
  if (ragged) {
    HK(hipMalloc((void **)&d_off, (n_reads + 1) * 8));
    HK(hipMemcpy(d_off, off.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice));
  }

  // ---- correctness: one submit on a fresh accumulator vs a host count ----
  if (!skip_check) {
    qk_accum *acc;
    CK(qk_accum_create(&acc, 0, adapters ? bits.data() : nullptr, max_len));
    qk_debug_set_mode(0);
    CK(qk_accum_submit_device(acc, d_seq, d_qual, d_off, n_reads, total, max_len, nullptr));
    std::vector<qk_base_info> got(max_len);
    uint64_t ml, nr;
    CK(qk_accum_finish(acc, got.data(), max_len, &ml, &nr));
    std::vector<qk_base_info> want(max_len);
    memset(want.data(), 0, max_len * sizeof(qk_base_info));
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t r = 0; r < n_reads; ++r) {
      const uint8_t *s = &seq[off[r]], *q = &qual[off[r]];
      uint32_t l = (uint32_t)(off[r + 1] - off[r]);
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

  // ---- sweep -------------------------------------------------------------
  struct Cfg { int T, U, tile, wgs; };
  std::vector<Cfg> cfgs = {
      {1024, 4, 192, 2}, {1024, 2, 192, 2}, {1024, 1, 192, 2}, {512, 4, 192, 4}, {512, 2, 192, 4},
      {256, 4, 192, 8},  {1024, 4, 192, 1}, {1024, 4, 192, 4}, {512, 4, 192, 8}, {1024, 4, 304, 1},
      {512, 4, 304, 2},  {1024, 4, 96, 2},
  };
  const int modes_fixed[] = {0, 1, 2, 3};
  const double alg_bytes = 2.0 * total + (ragged ? 8.0 * n_reads : 0.0);
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
      int rc = qk_accum_submit_device(acc, d_seq, d_qual, d_off, n_reads, total, max_len, nullptr);
      if (rc) {
        printf("T=%4d U=%d tile=%3d wgs=%d mode=%d : skipped (%s)\n", c.T, c.U, c.tile, c.wgs, mode, qk_last_error());
        qk_accum_destroy(acc);
        continue;
      }
      CK(qk_accum_submit_device(acc, d_seq, d_qual, d_off, n_reads, total, max_len, nullptr));
      CK(qk_accum_sync(acc));
      CK(qk_accum_timing_enable(acc, 1));
      const int iters = 10;
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters; ++i)
        CK(qk_accum_submit_device(acc, d_seq, d_qual, d_off, n_reads, total, max_len, nullptr));
      CK(qk_accum_sync(acc));
      double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters;
      double ms;
      uint64_t launches;
      CK(qk_accum_timing_read(acc, &ms, &launches));
      double k = ms / launches * 1e-3;
      printf("T=%4d U=%d tile=%3d wgs=%d mode=%d : hist kernel %.3f ms  %.2f TB/s (%.1f%% of 8)  %.1f Gbases/s | wall/step %.3f ms\n",
             c.T, c.U, c.tile, c.wgs, mode, k * 1e3, alg_bytes / k / 1e12, alg_bytes / k / 8e12 * 100,
             total / k / 1e9, wall * 1e3);
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
