// tools/sdwa_ds_rate.hip — developer tool (round 5): what it costs to rebuild the address of an LDS atomic IN PLACE.
// The wide histogram layout (qk_kernels.hip.h, qhist_index_wide) wants  v_and_b32_sdwa addr, 0x7f, w dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE
// followed by  ds_add_u32 addr, one  — one VALU instruction per base — and the first build of it ran the kernel at half speed.
// This measures the pair with N address registers in rotation (the distance, in LDS instructions, between an atomic that reads a
// register and the SDWA instruction that rewrites it), against the two-instruction address of the narrow layout.
//   hipcc --offload-arch=gfx950 -O2 -o tools/sdwa_ds_rate tools/sdwa_ds_rate.hip && tools/sdwa_ds_rate
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: narrow (v_lshrrev tmp; v_bitop3 tmp; ds_add tmp)   1: sdwa preserve, N registers in rotation
// MODE 2: sdwa into a fresh temporary (v_mov tmp, addr; sdwa tmp; ds_add tmp)   3: sdwa preserve + s_nop 4 in front of it
template <int MODE, int N>
__global__ __launch_bounds__(1024) void k(unsigned *out, unsigned seed) {
  extern __shared__ unsigned lds[];
  for (unsigned i = threadIdx.x; i < 32768; i += 1024) lds[i] = 0;
  __syncthreads();
  unsigned addr[16];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (int i = 0; i < 16; ++i) addr[i] = ((lane + 4u * i) & 63u) * 4u;   // column = byte 0; two lanes per bank
  unsigned w = seed * 2654435761u + threadIdx.x * 40503u, m7f = 0x7fu, m3f80 = 0x3f80u, one = 1u;
  (void)wave;
  for (int it = 0; it < 1000; ++it) {
    w = w * 1664525u + 1013904223u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MODE == 0) {
        unsigned t;
        asm volatile("v_lshrrev_b32 %0, 1, %1\n v_bitop3_b32 %0, %0, %2, %3 bitop3:0xea\n ds_add_u32 %0, %4" : "=&v"(t) : "v"(w), "v"(m3f80), "v"(addr[r % N]), "v"(one) : "memory");
      } else if (MODE == 1) {
        asm volatile("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2\n ds_add_u32 %0, %3"
                     : "+v"(addr[r % N]) : "v"(m7f), "v"(w), "v"(one) : "memory");
      } else if (MODE == 2) {
        unsigned t;
        asm volatile("v_mov_b32 %0, %1\n v_and_b32_sdwa %0, %2, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2\n ds_add_u32 %0, %4"
                     : "=&v"(t) : "v"(addr[r % N]), "v"(m7f), "v"(w), "v"(one) : "memory");
      } else {
        asm volatile("s_nop 4\n v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2\n ds_add_u32 %0, %3"
                     : "+v"(addr[r % N]) : "v"(m7f), "v"(w), "v"(one) : "memory");
      }
    }
  }
  __syncthreads();
  unsigned s = 0;
  for (unsigned i = threadIdx.x; i < 32768; i += 1024) s += lds[i];
  if (s == 0x12345678u) out[0] = s;
}

template <int MODE, int N>
static double run(unsigned *d) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void *)k<MODE, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL((k<MODE, N>), dim3(256), dim3(1024), 131072, 0, d, 3u);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, N>), dim3(256), dim3(1024), 131072, 0, d, 5u);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  unsigned *d;
  (void)hipMalloc((void **)&d, 4);
#define R(M, N, what) printf("%-64s %8.3f ms\n", what, run<M, N>(d));
  R(0, 16, "narrow: v_lshrrev + v_bitop3 + ds_add (fresh temporary)");
  R(1, 1, "sdwa preserve + ds_add, 1 address register");
  R(1, 2, "sdwa preserve + ds_add, 2 address registers in rotation");
  R(1, 4, "sdwa preserve + ds_add, 4");
  R(1, 8, "sdwa preserve + ds_add, 8");
  R(1, 16, "sdwa preserve + ds_add, 16");
  R(2, 16, "v_mov + sdwa preserve on the copy + ds_add");
  R(3, 4, "s_nop 4 + sdwa preserve + ds_add, 4");
  return 0;
}
