// atomic_rate — developer tool: what the histogram kernels' flush pays for its global atomics on gfx950.
// Every workgroup (1024 threads, one per CU) adds into rows of a planar table the way flush() does: a
// wave-instruction covers 64 consecutive counters of one row; rows are `stride` counters apart; the workgroups
// of a group of G share a tile (the same rows and columns), different groups use different column ranges.
// Compared: 64-bit and 32-bit counters, no-return atomics.  Prints us per flush-sized burst and bytes/s.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/atomic_rate tools/atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <typename T>
__global__ __launch_bounds__(1024) void burst(T *table, unsigned stride, unsigned rows, unsigned cols, unsigned group, int reps) {
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const unsigned col0 = (blockIdx.x / group) * cols;
  for (int r = 0; r < reps; ++r)
    for (unsigned row = wave; row < rows; row += 16u)
      for (unsigned c = lane; c < cols; c += 64u) atomicAdd(&table[(size_t)row * stride + col0 + c], (T)1);
}

template <typename T>
static void run(const char *name, unsigned rows, unsigned cols, unsigned group, unsigned stride) {
  T *d;
  const size_t n = (size_t)97 * stride;
  hipMalloc(&d, n * sizeof(T));
  hipMemset(d, 0, n * sizeof(T));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int reps = 8;
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(a);
    burst<T><<<256, 1024>>>(d, stride, rows, cols, group, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double per = ms * 1e3 / reps, bytes = (double)rows * cols * sizeof(T) * 256;
  printf("%-4s rows %3u cols %4u, %u workgroups per tile: %7.1f us per burst, %6.0f GB/s of added bytes, %5.1f ns per wave-instruction per CU\n",
         name, rows, cols, group, per, bytes / (per * 1e-6) / 1e9, per * 1e3 / (rows * cols / 64.0));
  hipFree(d);
}

int main() {
  for (unsigned group : {1u, 7u}) {
    run<unsigned long long>("u64", 40, 512, group, 20000);
    run<unsigned>("u32", 40, 512, group, 20000);
    run<unsigned long long>("u64", 91, 512, group, 20000);
    run<unsigned>("u32", 91, 512, group, 20000);
    run<unsigned long long>("u64", 40, 152, group, 152);
    run<unsigned>("u32", 40, 152, group, 152);
  }
  return 0;
}
