// piece_rate — developer microbenchmark (not part of the product): what HBM delivers for the LONG-READ access
// pattern of qk::hist_kernel, loads only.  A workgroup of 16 waves works on ONE position tile of many reads: per
// step every wave fetches one read's piece of the tile (64 lanes x W bytes, contiguous, 128-byte aligned) from both
// arrays; consecutive reads lie ~10 kb apart.  Questions: (1) what does the piece size (W = 8 -> 512-byte pieces,
// the product's; 16 -> 1 KiB; 32 -> 2 KiB) cost against a contiguous stream, (2) does it matter which workgroups
// run next to each other — tile-major shares (the product's static split: neighbours work on the SAME tile of
// DIFFERENT reads) or read-major items (neighbours work on NEIGHBOURING tiles of the SAME reads, so that a read's
// DRAM pages are fetched by several workgroups at about the same time).
//   hipcc -O3 --offload-arch=gfx950 -o tools/piece_rate tools/piece_rate.hip && tools/piece_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define HK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "HIP FAIL %s: %s\n", #x, hipGetErrorString(e_));        \
      exit(2);                                                                \
    }                                                                         \
  } while (0)

struct Item { uint32_t tile, r0, r1; };   // reads order[r0 .. r1) x one tile

// W bytes per lane and array; PD-deep software pipeline of steps of 16 reads (one per wave) x U
// LPR lanes per read-piece (64: a wave per read; 32: two reads per wave), K loads of W bytes per lane and array,
// LPR*W bytes apart: a read's piece of the tile is K * LPR * W bytes
template <int W, int U, int LPR = 64, int K = 1>
__global__ __launch_bounds__(1024) void piece_kernel(const uint8_t *a, const uint8_t *b, const unsigned long long *starts,
                                                     const uint32_t *lens, const uint32_t *order, const Item *items,
                                                     const uint32_t *first_item, unsigned *sink) {
  const uint32_t wave = threadIdx.x / LPR, lane = threadIdx.x % LPR;
  constexpr uint32_t TILE = (uint32_t)LPR * W * K, RPS = 1024 / LPR;   // reads per step and unroll
  uint32_t acc = 0;
  __shared__ unsigned long long offs[8192];   // staged like the product does: byte offset of every read's piece of this tile
  for (uint32_t it = first_item[blockIdx.x]; it < first_item[blockIdx.x + 1]; ++it) {
    const Item im = items[it];
    const uint32_t pos = im.tile * TILE + lane * W;
    typedef uint32_t vec __attribute__((ext_vector_type(W / 4)));
    for (uint32_t p0 = im.r0; p0 < im.r1; p0 += 8192u) {
      const uint32_t p1 = p0 + 8192u < im.r1 ? p0 + 8192u : im.r1;
      __syncthreads();
      for (uint32_t i = p0 + threadIdx.x; i < p1; i += 1024u) {
        const uint32_t id = order[i];
        offs[i - p0] = starts[id] + (pos - lane * W < lens[id] ? im.tile * TILE : 0u);
      }
      __syncthreads();
      vec x[2][U * K], y[2][U * K];
      auto issue = [&](uint32_t r, int slot) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t rr = r + (uint32_t)u * RPS + wave;
          const unsigned long long off = offs[rr < p1 ? rr - p0 : 0u] + lane * W;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            x[slot][u * K + k] = *reinterpret_cast<const vec *>(a + off + (size_t)k * LPR * W);
            y[slot][u * K + k] = *reinterpret_cast<const vec *>(b + off + (size_t)k * LPR * W);
          }
        }
      };
      issue(p0, 0);
      int cur = 0;
      for (uint32_t r = p0; r < p1; r += RPS * U) {
        issue(r + RPS * U, cur ^ 1);
#pragma unroll
        for (int u = 0; u < U * K; ++u)
#pragma unroll
          for (int k = 0; k < W / 4; ++k) acc ^= x[cur][u][k] ^ y[cur][u][k];
        cur ^= 1;
      }
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int U>
__global__ __launch_bounds__(1024) void stream_kernel(const uint4 *a, const uint4 *b, size_t n16, unsigned *sink) {
  uint32_t acc = 0;
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
  for (size_t i = lo + threadIdx.x; i + 1024 * (U - 1) < hi; i += 1024 * U) {
    uint4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { x[u] = a[i + 1024 * u]; y[u] = b[i + 1024 * u]; }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ x[u].z ^ x[u].w ^ y[u].x ^ y[u].y ^ y[u].z ^ y[u].w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// fixed-length reads of L bytes (L % 4 == 0), contiguous: lane (read row, chunk) loads W bytes at read * L + chunk * W
// — 4-byte aligned, not W-byte aligned (300-base reads: the histogram kernel's dword-aligned fixed-length path)
template <int W>
__global__ __launch_bounds__(1024) void fixed_kernel(const uint8_t *a, const uint8_t *b, uint32_t L, uint32_t n_reads, unsigned *sink) {
  typedef uint32_t vec __attribute__((ext_vector_type(W / 4)));
  const uint32_t cpr = (L + W - 1) / W;          // chunks (lanes) per read
  const uint32_t rw = 1024u / cpr;                // reads per step
  const uint32_t ri = threadIdx.x / cpr, ch = threadIdx.x % cpr;
  const uint32_t per = (n_reads + gridDim.x - 1) / gridDim.x;
  const uint32_t r0 = blockIdx.x * per, r1 = r0 + per < n_reads ? r0 + per : n_reads;
  uint32_t acc = 0;
  if (ri < rw) {
    vec x[2], y[2];
    auto issue = [&](uint32_t r, int slot) {
      const uint32_t rr = r + ri < r1 ? r + ri : r0;
      const size_t off = (size_t)rr * L + ch * W;
      x[slot] = *reinterpret_cast<const vec *>(__builtin_assume_aligned(a + off, 4));
      y[slot] = *reinterpret_cast<const vec *>(__builtin_assume_aligned(b + off, 4));
    };
    issue(r0, 0);
    int cur = 0;
    for (uint32_t r = r0; r < r1; r += rw) {
      issue(r + rw, cur ^ 1);
#pragma unroll
      for (int k = 0; k < W / 4; ++k) acc ^= x[cur][k] ^ y[cur][k];
      cur ^= 1;
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 143000u;
  const uint32_t lo = argc > 2 ? (uint32_t)atoi(argv[2]) : 1000u, hi = argc > 3 ? (uint32_t)atoi(argv[3]) : 20000u;
  const int grid = 256;
  std::vector<uint32_t> lens(n);
  std::vector<unsigned long long> starts(n);
  uint64_t s = 6, total = 0, bases = 0;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
  for (uint32_t r = 0; r < n; ++r) {
    lens[r] = lo + rnd() % (hi - lo + 1);
    starts[r] = total;
    total += (lens[r] + 127u) / 128u * 128u;
    bases += lens[r];
  }
  uint8_t *a, *b;
  unsigned *sink;
  HK(hipMalloc((void **)&a, total + 4096));
  HK(hipMalloc((void **)&b, total + 4096));
  HK(hipMemset(a, 1, total + 4096));
  HK(hipMemset(b, 2, total + 4096));
  HK(hipMalloc((void **)&sink, 4));
  unsigned long long *d_starts;
  uint32_t *d_lens, *d_order;
  HK(hipMalloc((void **)&d_starts, n * 8));
  HK(hipMalloc((void **)&d_lens, n * 4));
  HK(hipMalloc((void **)&d_order, n * 4));
  HK(hipMemcpy(d_starts, starts.data(), n * 8, hipMemcpyHostToDevice));
  HK(hipMemcpy(d_lens, lens.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0));
  HK(hipEventCreate(&e1));
  printf("%u reads of %u..%u bases on 128-byte lines: %.2f GB per array, %.2f Gbases\n", n, lo, hi, total / 1e9, bases / 1e9);

  {   // the ceiling: both arrays as one contiguous stream
    for (int rep = 0; rep < 2; ++rep) {
      HK(hipEventRecord(e0, 0));
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream_kernel<4>, dim3(grid), dim3(1024), 0, 0, (const uint4 *)a, (const uint4 *)b, total / 16, sink);
      HK(hipEventRecord(e1, 0));
      HK(hipEventSynchronize(e1));
      float ms;
      HK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("  contiguous stream, 16 B/lane, slice per workgroup            : %.3f ms  %.2f TB/s\n", ms / 10, 2.0 * total / (ms / 10 * 1e-3) / 1e12);
    }
  }

  for (uint32_t L : {300u, 304u, 152u, 100u}) {
    const uint32_t nr = (uint32_t)(total / L);
    for (int W : {8, 16}) {
      for (int rep = 0; rep < 2; ++rep) {
        HK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) {
          if (W == 8) hipLaunchKernelGGL(fixed_kernel<8>, dim3(grid), dim3(1024), 0, 0, a, b, L, nr, sink);
          else hipLaunchKernelGGL(fixed_kernel<16>, dim3(grid), dim3(1024), 0, 0, a, b, L, nr, sink);
        }
        HK(hipEventRecord(e1, 0));
        HK(hipEventSynchronize(e1));
        float ms;
        HK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("  fixed-length reads of %3u bytes, %2d B per lane (4-byte aligned), one read step in flight: %.3f ms  %.2f TB/s\n", L, W, ms / 10,
                        2.0 * nr * L / (ms / 10 * 1e-3) / 1e12);
      }
    }
  }

  auto run = [&](int W, bool sorted, int layout, uint32_t chunk, int lpr = 64, int kk = 1) {
    // layout 0: tile-major equal shares (product) | 1: read-major items of `chunk` reads, dealt round-robin |
    // 2: read-major items, contiguous runs of items per workgroup
    const uint32_t TILE = (uint32_t)lpr * (uint32_t)W * (uint32_t)kk;
    const uint32_t n_tiles = (hi + TILE - 1) / TILE;
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    if (sorted) std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return (lens[x] + TILE - 1) / TILE > (lens[y] + TILE - 1) / TILE; });
    std::vector<uint32_t> reach(n_tiles);
    for (uint32_t t = 0; t < n_tiles; ++t) {
      uint32_t c = 0;
      if (sorted) {
        uint32_t l = 0, h = n;   // order is descending in tiles reached: binary search
        while (l < h) { uint32_t m = (l + h) / 2; if (lens[order[m]] > t * TILE) l = m + 1; else h = m; }
        c = l;
      } else c = n;
      reach[t] = c;
    }
    std::vector<Item> items;
    std::vector<uint32_t> first(grid + 1, 0);
    uint64_t W_total = 0;
    for (auto c : reach) W_total += c;
    uint64_t piece_bytes = 0;
    for (uint32_t t = 0; t < n_tiles; ++t)
      for (uint32_t i = 0; i < reach[t]; ++i) {
        const uint32_t l = lens[order[i]];
        if (l > t * TILE) piece_bytes += std::min<uint32_t>(TILE, (l - t * TILE + 127u) / 128u * 128u);
      }
    if (layout == 0) {
      std::vector<uint64_t> prefix(n_tiles + 1, 0);
      for (uint32_t t = 0; t < n_tiles; ++t) prefix[t + 1] = prefix[t] + reach[t];
      for (int g = 0; g < grid; ++g) {
        uint64_t l = W_total * g / grid, h = W_total * (g + 1) / grid;
        first[g] = (uint32_t)items.size();
        for (uint32_t t = 0; t < n_tiles && l < h; ++t) {
          if (prefix[t + 1] <= l) continue;
          const uint64_t sh = std::min<uint64_t>(h, prefix[t + 1]);
          items.push_back({t, (uint32_t)(l - prefix[t]), (uint32_t)(sh - prefix[t])});
          l = sh;
        }
      }
      first[grid] = (uint32_t)items.size();
    } else {
      std::vector<Item> all;
      for (uint32_t j = 0; j * chunk < n; ++j)
        for (uint32_t t = 0; t < n_tiles; ++t)
          if (reach[t] > j * chunk) all.push_back({t, j * chunk, std::min(reach[t], (j + 1) * chunk)});
      if (layout == 1) {   // item k -> workgroup k % grid (what a dynamic queue gives, roughly)
        for (int g = 0; g < grid; ++g) {
          first[g] = (uint32_t)items.size();
          for (size_t k = g; k < all.size(); k += grid) items.push_back(all[k]);
        }
      } else {
        for (int g = 0; g < grid; ++g) {
          first[g] = (uint32_t)items.size();
          for (size_t k = all.size() * g / grid; k < all.size() * (g + 1) / grid; ++k) items.push_back(all[k]);
        }
      }
      first[grid] = (uint32_t)items.size();
    }
    Item *d_items;
    uint32_t *d_first;
    HK(hipMalloc((void **)&d_items, items.size() * sizeof(Item)));
    HK(hipMalloc((void **)&d_first, (grid + 1) * 4));
    HK(hipMemcpy(d_items, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice));
    HK(hipMemcpy(d_first, first.data(), (grid + 1) * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_order, order.data(), n * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
      HK(hipEventRecord(e0, 0));
      for (int i = 0; i < 10; ++i) {
#define ARGS dim3(grid), dim3(1024), 0, 0, a, b, d_starts, d_lens, d_order, d_items, d_first, sink
        if (W == 8 && lpr == 64 && kk == 1) hipLaunchKernelGGL((piece_kernel<8, 2>), ARGS);
        else if (W == 8 && lpr == 64 && kk == 2) hipLaunchKernelGGL((piece_kernel<8, 1, 64, 2>), ARGS);
        else if (W == 8 && lpr == 64 && kk == 4) hipLaunchKernelGGL((piece_kernel<8, 1, 64, 4>), ARGS);
        else if (W == 16 && lpr == 64) hipLaunchKernelGGL((piece_kernel<16, 2>), ARGS);
        else if (W == 16 && lpr == 32 && kk == 1) hipLaunchKernelGGL((piece_kernel<16, 2, 32, 1>), ARGS);
        else if (W == 16 && lpr == 32 && kk == 2) hipLaunchKernelGGL((piece_kernel<16, 1, 32, 2>), ARGS);
        else if (W == 8 && lpr == 32) hipLaunchKernelGGL((piece_kernel<8, 2, 32, 1>), ARGS);
        else if (W == 32) hipLaunchKernelGGL((piece_kernel<32, 1>), ARGS);
        else { printf("variant not built\n"); return; }
      }
      HK(hipEventRecord(e1, 0));
      HK(hipEventSynchronize(e1));
      float ms;
      HK(hipEventElapsedTime(&ms, e0, e1));
      if (rep)
        printf("  %4u-byte pieces (%2d B x %d per lane, %d lanes per read), %s, %-30s %5zu items: %.3f ms  %.2f TB/s of pieces (%.2f of bases)\n", TILE, W, kk, lpr, sorted ? "sorted" : "natural",
               layout == 0 ? "tile-major equal shares" : (layout == 1 ? "read-major items, round-robin" : "read-major items, runs"), items.size(),
               ms / 10, 2.0 * piece_bytes / (ms / 10 * 1e-3) / 1e12, 4.0 * bases / 2 / (ms / 10 * 1e-3) / 1e12);
    }
    HK(hipFree(d_items));
    HK(hipFree(d_first));
  };
  run(8, true, 0, 0);                 // the product: 512-byte pieces, 8 B per lane
  run(16, true, 0, 0, 32, 1);         // 512-byte pieces, 16 B per lane on half a wave (a wave holds two reads' pieces)
  run(8, true, 0, 0, 32, 1);          // 256-byte pieces, 8 B per lane
  run(8, true, 0, 0, 64, 2);          // 1 KiB pieces as two 8-byte loads per lane
  run(8, true, 0, 0, 64, 4);          // 2 KiB pieces as four
  run(16, true, 0, 0);                // 1 KiB pieces, 16 B per lane
  run(16, true, 0, 0, 32, 2);         // 1 KiB pieces, 2 x 16 B on half a wave
  run(32, true, 0, 0);                // 2 KiB pieces, 32 B per lane
  run(8, true, 1, 4096);              // read-major items instead of tile-major shares
  run(16, true, 1, 4096);
  return 0;
}
