// qk_adapter_kernels.hip.h — adapter 10-mer "first hit" path
// (reference: quack.c:206-217, table built by read_adapters quack.c:154-178).
//
// Reference semantics, closed form: let e* be the smallest e in [9, l-1] such
// that the 10-mer s[e-9..e] is in the table; the serial scan stops with
// i = e*+1 (i = l when there is none) and counts bases[i].kmer_count++ iff
// i < l.  The parallel form tests every window independently and takes a
// per-read minimum:
//   scan kernel : lane owns 16 positions, rebuilds the 2-bit codes of those and
//                 the 9 preceding bases, tests each window against a 32 KiB
//                 LDS-resident pre-filter (hash = low 18 bits of the 20-bit
//                 index) and, on a filter hit only, the exact 2^20-bit table
//                 in global memory (L2 resident); atomicMin(first_hit[read]).
//   count kernel: one thread per read turns first_hit into the kmer_count
//                 increment, privatised in LDS for positions < 4096.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "qk_kernels.hip.h"

namespace qk {

constexpr uint32_t kFilterBits = 1u << kFusedFilterLog2;  // 32 KiB of LDS, shared with the fused path
constexpr int kScanThreads = 256;
constexpr int kCountThreads = 256;
constexpr uint32_t kCountLdsPos = 4096;

// base byte -> 2-bit code, quack.c:148-150,201 (letter key c & 31: T=20, C=3, G=7, else A)
__device__ __forceinline__ uint32_t base_code(uint32_t c) {
  const uint32_t k = c & 31u;
  return k == 20u ? 1u : (k == 3u ? 2u : (k == 7u ? 3u : 0u));
}

__device__ __forceinline__ uint4 load16(const uint8_t *p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

template <bool FIXED>
__global__ __launch_bounds__(kScanThreads) void adapter_scan_kernel(const HistParams p, uint32_t lpr,
                                                                    uint32_t max_chunks) {
  __shared__ uint32_t filt[kFilterBits / 32];
  for (uint32_t i = threadIdx.x; i < kFilterBits / 32; i += kScanThreads) filt[i] = p.kmer_filter[i];
  __syncthreads();

  const uint32_t ri = threadIdx.x / lpr;
  const uint32_t sub = threadIdx.x - ri * lpr;
  const uint32_t rw = kScanThreads / lpr;
  if (ri >= rw) return;
  for (uint64_t r = (uint64_t)blockIdx.x * rw + ri; r < p.n_reads; r += (uint64_t)gridDim.x * rw) {
    uint64_t start;
    uint32_t len;
    if (FIXED) {
      start = r * (uint64_t)p.stride;
      len = p.lengths ? p.lengths[r] : p.read_len;   // (strided batch: fixed stride, own lengths)
    } else {
      start = p.offsets[r];
      len = p.lengths ? p.lengths[r] : (uint32_t)(p.offsets[r + 1] - start);
    }
    if (len <= 10u) continue;  // i = 10 >= l: nothing can be counted
    const uint8_t *s = p.seq + start;
    for (uint32_t c = sub; c < max_chunks; c += lpr) {
      const uint32_t cpos = c * 16u;
      if (cpos >= len) break;
      // codes of positions cpos-9 .. cpos+15, earliest base most significant
      uint32_t own[4], prev[4] = {0, 0, 0, 0};
      {
        uint4 v = load16(s + cpos);
        own[0] = v.x; own[1] = v.y; own[2] = v.z; own[3] = v.w;
        if (cpos >= 16u) {
          uint4 w = load16(s + cpos - 16u);
          prev[0] = w.x; prev[1] = w.y; prev[2] = w.z; prev[3] = w.w;
        }
      }
      uint64_t packed = 0;
#pragma unroll
      for (int t = 7; t < 16; ++t)  // positions cpos-9 .. cpos-1
        packed = (packed << 2) | base_code((prev[t >> 2] >> (8 * (t & 3))) & 0xFFu);
#pragma unroll
      for (int t = 0; t < 16; ++t)
        packed = (packed << 2) | base_code((own[t >> 2] >> (8 * (t & 3))) & 0xFFu);
      // window ending at own position j covers packed bits [2*(15-j), 2*(15-j)+20)
      uint32_t best = kNoHit;
#pragma unroll
      for (int j = 15; j >= 0; --j) {
        const uint32_t e = cpos + (uint32_t)j;
        const uint32_t km = (uint32_t)(packed >> (2 * (15 - j))) & 0xFFFFFu;
        const uint32_t h = km & (kFilterBits - 1u);
        const bool maybe = (filt[h >> 5] >> (h & 31u)) & 1u;
        if (maybe && e >= 9u && e < len) {
          if ((p.kmer_bits[km >> 5] >> (km & 31u)) & 1u) best = e;
        }
      }
      if (best != kNoHit) atomicMin(&p.first_hit[r], best);
    }
  }
}

template <bool FIXED>
__global__ __launch_bounds__(kCountThreads) void adapter_count_kernel(const HistParams p) {
  __shared__ uint32_t cnt[kCountLdsPos];
  for (uint32_t i = threadIdx.x; i < kCountLdsPos; i += kCountThreads) cnt[i] = 0;
  __syncthreads();
  for (uint64_t r = (uint64_t)blockIdx.x * kCountThreads + threadIdx.x; r < p.n_reads;
       r += (uint64_t)gridDim.x * kCountThreads) {
    const uint32_t fh = p.first_hit[r];
    if (fh == kNoHit) continue;
    uint32_t len;
    if (FIXED) len = p.read_len;
    else len = p.lengths ? p.lengths[r] : (uint32_t)(p.offsets[r + 1] - p.offsets[r]);
    const uint32_t i = fh + 1u;      // quack.c:211-213: i ends one past the window
    if (i < len) {                   // quack.c:215
      if (i < kCountLdsPos) atomicAdd(&cnt[i], 1u);
      else atomicAdd(&p.table[(uint64_t)kRowKmer * p.table_len + i], 1ull);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kCountLdsPos; i += kCountThreads) {
    const uint32_t c = cnt[i];
    if (c != 0 && i < p.table_len)
      atomicAdd(&p.table[(uint64_t)kRowKmer * p.table_len + i], (unsigned long long)c);
  }
}

inline int launch_adapter_count(const HistParams &hp, int n_cu, hipStream_t st) {
  if (hp.n_reads == 0) return 0;
  uint64_t cblocks = (hp.n_reads + kCountThreads - 1) / kCountThreads;
  if (cblocks > (uint64_t)n_cu * 4) cblocks = (uint64_t)n_cu * 4;
  if (hp.offsets == nullptr && hp.lengths == nullptr)
    hipLaunchKernelGGL(adapter_count_kernel<true>, dim3((unsigned)cblocks), dim3(kCountThreads), 0, st, hp);
  else
    hipLaunchKernelGGL(adapter_count_kernel<false>, dim3((unsigned)cblocks), dim3(kCountThreads), 0, st, hp);
  return (int)hipGetLastError();
}

// returns a hipError_t as int
inline int launch_adapter_scan(const HistParams &hp, int n_cu, hipStream_t st) {
  if (hp.n_reads == 0) return 0;
  hipError_t e = hipMemsetAsync(hp.first_hit, 0xFF, hp.n_reads * sizeof(uint32_t), st);
  if (e != hipSuccess) return (int)e;
  const bool fixed = hp.offsets == nullptr;
  const uint32_t max_len = fixed ? hp.read_len : hp.n_tiles * hp.tile_pos;
  const uint32_t max_chunks = (max_len + 15u) / 16u;
  uint32_t lpr = max_chunks < 1u ? 1u : max_chunks;
  if (lpr > (uint32_t)kScanThreads) lpr = kScanThreads;
  const uint32_t rw = kScanThreads / lpr;
  uint64_t blocks = (hp.n_reads + rw - 1) / rw;
  const uint64_t cap = (uint64_t)n_cu * 8;
  if (blocks > cap) blocks = cap;
  if (fixed)
    hipLaunchKernelGGL(adapter_scan_kernel<true>, dim3((unsigned)blocks), dim3(kScanThreads), 0, st, hp, lpr, max_chunks);
  else
    hipLaunchKernelGGL(adapter_scan_kernel<false>, dim3((unsigned)blocks), dim3(kScanThreads), 0, st, hp, lpr, max_chunks);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return launch_adapter_count(hp, n_cu, st);
}

// Build the LDS-resident exact table of the fused path (the queued candidates are
// checked against it without leaving the LDS): 2^log2b buckets of
// eight u16 slots, keyed by km*mul mod 2^20.  Returns false when no (mul, size)
// up to `max_log2` avoids a bucket overflow (very large adapter sets): the
// kernel then falls back to the global bitset for filter hits.
inline bool build_kmer_buckets(const uint32_t *host_bits, uint32_t max_log2, std::vector<uint16_t> *out,
                               uint32_t *log2b_out, uint32_t *mul_out) {
  std::vector<uint32_t> kms;
  for (uint32_t w = 0; w < (1u << 15); ++w) {
    uint32_t v = host_bits[w];
    while (v) {
      kms.push_back(w * 32u + (uint32_t)__builtin_ctz(v));
      v &= v - 1;
    }
  }
  uint32_t log2b = 6;
  while (log2b < max_log2 && (8u << log2b) < kms.size() * 3) ++log2b;   // ~1/3 full to start with
  uint64_t seed = 0x9E3779B97F4A7C15ull;
  for (; log2b <= max_log2; ++log2b) {
    for (int attempt = 0; attempt < 32; ++attempt) {
      seed = seed * 6364136223846793005ull + 1442695040888963407ull;
      const uint32_t mul = ((uint32_t)(seed >> 33) & 0xFFFFFu) | 1u;
      std::vector<uint16_t> tab((size_t)8 << log2b, 0);
      std::vector<uint8_t> fill((size_t)1 << log2b, 0);
      bool ok = true;
      for (uint32_t km : kms) {
        const uint32_t h = (km * mul) & 0xFFFFFu;
        const uint32_t b = h >> (20 - log2b);
        if (fill[b] == 8) {
          ok = false;
          break;
        }
        tab[(size_t)b * 8 + fill[b]++] = (uint16_t)((h & ((1u << (20 - log2b)) - 1u)) | 0x8000u);
      }
      if (ok) {
        *out = tab;
        *log2b_out = log2b;
        *mul_out = mul;
        return true;
      }
    }
  }
  return false;
}

// Upload the exact bitset and derive the LDS filters from it:
//   words [0, 2^13)      separate scan kernel: 2^18 bits keyed by a window's low 18 bits
//   words [2^13, 2^14)   fused path: 2^18 bits holding, for every adapter 10-mer, its
//                        prefix 9-mer and its suffix 9-mer, over COMPLEMENTED codes
//                        (hist_kernel probes the one 9-mer of a window that ends on an
//                        even position; byte = key >> 3, bit = key & 7, as it reads it)
inline int upload_kmer_tables(const uint32_t *host_bits, uint32_t **d_bits, uint32_t **d_filter,
                              uint32_t *filter_bits) {
  const uint32_t words = 1u << 15;  // 2^20 bits
  uint32_t *filt = (uint32_t *)calloc(2u * kFusedFilterWords, sizeof(uint32_t));
  if (!filt) return (int)hipErrorOutOfMemory;
  for (uint32_t w = 0; w < words; ++w) {
    uint32_t v = host_bits[w];
    while (v) {
      const uint32_t b = (uint32_t)__builtin_ctz(v);
      v &= v - 1;
      const uint32_t km = w * 32u + b;
      const uint32_t h = km & (kFilterBits - 1u);
      filt[h >> 5] |= 1u << (h & 31u);
      const uint32_t suffix9 = (km & 0x3FFFFu) ^ 0x3FFFFu, prefix9 = (km >> 2) ^ 0x3FFFFu;
      filt[kFusedFilterWords + (suffix9 >> 5)] |= 1u << (suffix9 & 31u);
      filt[kFusedFilterWords + (prefix9 >> 5)] |= 1u << (prefix9 & 31u);
    }
  }
  hipError_t e = hipMalloc((void **)d_bits, words * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemcpy(*d_bits, host_bits, words * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void **)d_filter, 2u * kFusedFilterWords * 4);
  if (e == hipSuccess) e = hipMemcpy(*d_filter, filt, 2u * kFusedFilterWords * 4, hipMemcpyHostToDevice);
  free(filt);
  *filter_bits = kFilterBits;
  return (int)e;
}

}  // namespace qk
