// floor of a HIP process: start-up and teardown with nothing in between
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
int main(int argc, char **argv) {
  auto t0 = std::chrono::steady_clock::now();
  (void)hipSetDevice(0);
  (void)hipFree(0);
  size_t mb = argc > 1 ? atoi(argv[1]) : 0;
  void *h = nullptr, *d = nullptr;
  if (mb) {
    (void)hipHostMalloc(&h, mb << 20);
    (void)hipMalloc(&d, mb << 20);
  }
  auto t1 = std::chrono::steady_clock::now();
  fprintf(stderr, "init+alloc %.3f s\n", std::chrono::duration<double>(t1 - t0).count());
  if (argc > 2) _exit(0);
  return 0;
}
