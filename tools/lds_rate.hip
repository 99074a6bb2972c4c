// lds_rate — developer tool: what one LDS instruction of a 64-lane wave costs a CU of gfx950, with all 16
// waves of a 1024-thread workgroup issuing it back to back (the histogram kernels' situation), alone and
// mixed with VALU work.  Prints ns and shader-clock cycles per wave-instruction per CU.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/lds_rate tools/lds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(S, A) S(A, 0) S(A, 1) S(A, 2) S(A, 3) S(A, 4) S(A, 5) S(A, 6) S(A, 7) S(A, 8) S(A, 9) S(A, 10) S(A, 11) S(A, 12) S(A, 13) S(A, 14) S(A, 15)

// OP: 0 ds_add_u32 (same row, bank == lane), 1 ds_add_u32 rows differ per lane (bank == lane),
// 2 ds_read_b32, 3 ds_read_u8, 4 ds_write_b32, 5 ds_bpermute_b32, 6 ds_add_u32 2-way bank conflict,
// 7 ds_add_rtn_u32, 8 ds_add_u32 all lanes one address, 9 ds_add_u32 with 16 lanes active,
// 10 ds_read_b64, 11 ds_add_u32 lanes pairwise on one address (l, l+32)
// VALU: v_add_u32 instructions issued per LDS instruction (0, 2, 4, 8, 16)
template <int OP, int VALU>
__global__ __launch_bounds__(1024) void k(unsigned *out, unsigned seed, int iters) {
  extern __shared__ unsigned lds[];
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  for (unsigned i = tid; i < 16384u; i += 1024u) lds[i] = 0;
  __syncthreads();
  unsigned addr[4];
  for (int i = 0; i < 4; ++i) {
    unsigned row = (OP == 1 || OP == 11) ? ((tid * 2654435761u + i * 40503u + seed) >> 9) & 127u : (unsigned)i;
    unsigned col = lane & 31u;
    if (OP == 6) col = (lane & 15u) * 2u;            // 2 lanes of a 32-group per bank... (stride 2 dwords)
    if (OP == 8) { col = 0; row = i; }
    if (OP == 11) row = ((lane & 31u) * 2654435761u + i * 40503u + seed) >> 9 & 127u;   // lane l and l+32: same address
    addr[i] = (row << 7) | (col << 2);
    if (OP == 0 || OP == 2 || OP == 3 || OP == 4 || OP == 7 || OP == 9 || OP == 10) addr[i] = (i << 8) | (lane << 2);   // 64 consecutive dwords
    if (OP == 10) addr[i] = (i << 9) | (lane << 3);
    if (OP == 5) addr[i] = ((lane * 5u + i) & 63u) << 2;
  }
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = tid + i;
  unsigned one = 1u, acc = 0;
  unsigned long long acc2 = 0;
  if (OP == 9 && (lane & 3u)) one = 0;   // placeholder; exec is narrowed below
  for (int it = 0; it < iters; ++it) {
#define STEP(A, N)                                                                                              \
    if (OP == 0 || OP == 1 || OP == 6 || OP == 8 || OP == 11) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(addr[N & 3]), "v"(one), "i"(16384 * (N >> 2)) : "memory"); \
    if (OP == 9) { if ((lane & 3u) == 0) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(addr[N & 3]), "v"(one), "i"(16384 * (N >> 2)) : "memory"); } \
    if (OP == 2) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(acc) : "v"(addr[N & 3]), "i"(16384 * (N >> 2)) : "memory"); \
    if (OP == 3) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(acc) : "v"(addr[N & 3]), "i"(16384 * (N >> 2)) : "memory"); \
    if (OP == 10) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(acc2) : "v"(addr[N & 3]), "i"(16384 * (N >> 2)) : "memory"); \
    if (OP == 4) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr[N & 3]), "v"(one), "i"(16384 * (N >> 2)) : "memory"); \
    if (OP == 5) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(acc) : "v"(addr[N & 3]), "v"(one) : "memory"); \
    if (OP == 7) asm volatile("ds_add_rtn_u32 %0, %1, %2 offset:%3" : "=v"(acc) : "v"(addr[N & 3]), "v"(one), "i"(16384 * (N >> 2)) : "memory"); \
    _Pragma("unroll") for (int q = 0; q < VALU; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(one));
    REP16(STEP, 0)
#undef STEP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned r = acc ^ (unsigned)acc2 ^ (unsigned)(acc2 >> 32);
  for (int i = 0; i < 8; ++i) r ^= v[i];
  __syncthreads();
  r ^= lds[tid];
  if (r == 0x12345678u) out[0] = r;
}

static hipEvent_t e0, e1;
template <int OP, int VALU>
static double run(unsigned *d, int iters) {
  hipFuncSetAttribute((const void *)k<OP, VALU>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<OP, VALU>), dim3(256), dim3(1024), 65536, 0, d, 3u, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<OP, VALU>), dim3(256), dim3(1024), 65536, 0, d, 5u, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  unsigned *d;
  (void)hipMalloc((void **)&d, 4);
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  int khz = 0;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const int iters = 2000;
  const double wave_instr = 16.0 * iters * 16.0;   // per CU: 16 waves x 16 LDS instructions x iters
  printf("shader clock %.0f MHz; 256 workgroups x 1024 threads; per CU and wave-instruction:\n", khz / 1000.0);
#define R(OP, VALU, NAME)                                                                                     \
  {                                                                                                           \
    const double ms = run<OP, VALU>(d, iters);                                                                \
    printf("%-44s +%2d v_add  %7.2f ns  %6.1f cycles\n", NAME, VALU, ms * 1e6 / wave_instr, ms * 1e-3 * khz * 1e3 / wave_instr); \
  }
  R(0, 0, "ds_add_u32 consecutive dwords")
  R(1, 0, "ds_add_u32 random rows, bank == lane")
  R(11, 0, "ds_add_u32 lanes l, l+32 one address")
  R(6, 0, "ds_add_u32 2-way bank conflict")
  R(8, 0, "ds_add_u32 one address for all lanes")
  R(9, 0, "ds_add_u32 16 of 64 lanes")
  R(7, 0, "ds_add_rtn_u32")
  R(2, 0, "ds_read_b32")
  R(10, 0, "ds_read_b64")
  R(3, 0, "ds_read_u8")
  R(4, 0, "ds_write_b32")
  R(5, 0, "ds_bpermute_b32")
  R(1, 2, "ds_add_u32 random rows")
  R(1, 4, "ds_add_u32 random rows")
  R(1, 8, "ds_add_u32 random rows")
  R(1, 16, "ds_add_u32 random rows")
  R(3, 4, "ds_read_u8")
  R(3, 8, "ds_read_u8")
  return 0;
}
