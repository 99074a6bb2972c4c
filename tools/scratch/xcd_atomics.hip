// experiment: the end-of-kernel flush of 256 workgroups into one table — device-scope atomics (executed at the
// memory side on a multi-XCD part) against workgroup-scope atomics into a replica per XCD (executed in that XCD's L2).
// Checks the sums, times both.   hipcc -O3 --offload-arch=gfx950 -o /tmp/xcd_atomics tools/scratch/xcd_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRows = 44, kPos = 152, kWords = kRows * kPos;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xFu;
}

template <int MODE>   // 0: device scope, one table; 1: workgroup scope, replica of the XCD; 2: device scope, replica of the XCD;
                      // 3: one table, every workgroup starts at another row; 4: replica = blockIdx % 8; 5: replica = blockIdx % 32
__global__ __launch_bounds__(1024) void flush_kernel(unsigned long long *table, unsigned *seen) {
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) atomicOr(&seen[x], 1u);
  unsigned long long *t = (MODE == 0 || MODE == 3) ? table : MODE == 4 ? table + (size_t)(blockIdx.x & 7u) * kWords
                          : MODE == 5 ? table + (size_t)(blockIdx.x & 31u) * kWords : table + (size_t)x * kWords;
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  for (unsigned row0 = wave; row0 < kRows; row0 += 16)
    for (unsigned p = lane; p < kPos; p += 64) {
      const unsigned row = MODE == 3 ? (row0 + blockIdx.x * 7u) % kRows : row0;
      const unsigned long long c = 1 + ((row + p + blockIdx.x) & 3u);
      if (MODE == 1)
        __hip_atomic_fetch_add(&t[row * kPos + p], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else
        __hip_atomic_fetch_add(&t[row * kPos + p], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int MODE>
static void run(const char *name) {
  unsigned long long *d;
  unsigned *seen;
  const size_t words = (size_t)kWords * 32;
  (void)hipMalloc((void **)&d, words * 8);
  (void)hipMalloc((void **)&seen, 64);
  (void)hipMemset(d, 0, words * 8);
  (void)hipMemset(seen, 0, 64);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int launches = 200, grid = 256;
  hipLaunchKernelGGL(flush_kernel<MODE>, dim3(grid), dim3(1024), 0, 0, d, seen);
  (void)hipEventRecord(e0, 0);
  for (int i = 1; i < launches; i++) hipLaunchKernelGGL(flush_kernel<MODE>, dim3(grid), dim3(1024), 0, 0, d, seen);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(words);
  std::vector<unsigned> hs(16);
  (void)hipMemcpy(h.data(), d, words * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hs.data(), seen, 64, hipMemcpyDeviceToHost);
  // expected: sum over blocks of 1 + ((row + p + b) & 3), times launches, summed over the replicas
  size_t bad = 0;
  for (int row = 0; row < kRows; row++)
    for (int p = 0; p < kPos; p++) {
      unsigned long long want = 0, got = 0;
      for (int b = 0; b < grid; b++) want += 1 + ((row + p + b) & 3u);
      want *= launches;
      for (int r = 0; r < 32; r++) got += h[(size_t)r * kWords + row * kPos + p];
      bad += got != want;
    }
  int xcds = 0;
  for (int i = 0; i < 16; i++) xcds += hs[i] != 0;
  printf("%-44s %7.2f us per launch   %zu wrong sums   XCC ids seen: %d\n", name, 1e3 * ms / (launches - 1), bad, xcds);
  (void)hipFree(d);
  (void)hipFree(seen);
}

int main() {
  run<0>("device scope, one table");
  run<2>("device scope, a replica per XCD");
  run<1>("workgroup scope, a replica per XCD");
  run<0>("device scope, one table");
  run<1>("workgroup scope, a replica per XCD");
  run<3>("one table, rows rotated by the workgroup");
  run<4>("device scope, replica = blockIdx % 8");
  run<5>("device scope, replica = blockIdx % 32");
  run<3>("one table, rows rotated by the workgroup");
  return 0;
}
