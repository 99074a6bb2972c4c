// instr_rate — developer tool: issue cost of VALU instructions on gfx950 relative to v_add_u32, measured on
// four waves per SIMD, (a) as a chain of dependent instructions and (b) round-robin over eight registers.
// Why it exists: the fused adapter kernel is VALU-bound, and on this part the cost of an instruction
// depends on its encoding class (VOP2/VOP1 and v_bitop3 ~1.0, most other VOP3 ~1.55).
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X)                                                                                         \
  X(0, "v_add_u32", "v_add_u32 %0, %0, %1")                                                            \
  X(1, "v_and_b32", "v_and_b32 %0, %0, %1")                                                            \
  X(2, "v_xor_b32", "v_xor_b32 %0, %0, %1")                                                            \
  X(3, "v_lshrrev_b32 7", "v_lshrrev_b32 %0, 7, %0")                                                   \
  X(4, "v_lshlrev_b32 v", "v_lshlrev_b32 %0, %1, %0")                                                  \
  X(5, "v_min_u32", "v_min_u32 %0, %0, %1")                                                            \
  X(6, "v_mov_b32", "v_mov_b32 %0, %1")                                                                \
  X(7, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c")                                      \
  X(8, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 7")                                                          \
  X(9, "v_bfe_i32", "v_bfe_i32 %0, %0, %1, 1")                                                         \
  X(10, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2")                                               \
  X(11, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2")                                                 \
  X(12, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 3, %1")                                              \
  X(13, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 3, %1")                                                \
  X(14, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2")                                                     \
  X(15, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2")                                                       \
  X(16, "v_xad_u32", "v_xad_u32 %0, %0, %1, %2")                                                       \
  X(17, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2")                                                     \
  X(18, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, %2")                                             \
  X(19, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %1, %2")                                           \
  X(20, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %0, %1, %2")                                               \
  X(21, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1")                                                     \
  X(22, "v_mul_u32_u24 (VOP2)", "v_mul_u32_u24 %0, %0, %1")                                            \
  X(23, "v_mov_b32_dpp wave_shr:1", "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")      \
  X(24, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %1, %0")                                         \
  X(25, "v_cndmask_b32 (s[2:3])", "v_cndmask_b32_e64 %0, %0, %1, s[2:3]")                              \
  X(26, "v_add_u32 sdwa byte", "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
  X(27, "v_sub_u32", "v_sub_u32 %0, %0, %1")                                                           \
  X(28, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1")                                                     \
  X(29, "v_sad_u8", "v_sad_u8 %0, %0, %1, %2")                                                         \
  X(30, "v_lshrrev_b32 sgpr", "v_lshrrev_b32 %0, s4, %0")                                              \
  X(31, "v_lshlrev_b32 sgpr", "v_lshlrev_b32 %0, s4, %0")                                              \
  X(32, "v_lshlrev_b32 const", "v_lshlrev_b32 %0, 3, %0")                                              \
  X(33, "v_bitop3 (v, s, v)", "v_bitop3_b32 %0, %0, s5, %1 bitop3:0xe8")                               \
  X(34, "v_and_b32 literal", "v_and_b32 %0, 0x7f007f, %0")                                             \
  X(35, "v_or_b32", "v_or_b32 %0, %0, %1")                                                             \
  X(37, "v_cmp_ne_u32 + nothing", "v_cmp_ne_u32 vcc, %0, %1")                                          \
  X(38, "v_max_u32", "v_max_u32 %0, %0, %1")                                                           \
  X(39, "v_ashrrev_i32 const", "v_ashrrev_i32 %0, 3, %0")                                              \
  X(40, "v_lshl_add_u32 (shift 0)", "v_lshl_add_u32 %0, %0, 0, %1")                                    \
  X(41, "v_and sdwa BYTE_1 PRESERVE", "v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2") \
  X(42, "v_and sdwa BYTE_1 PAD", "v_and_b32_sdwa %0, %1, %0 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2") \
  X(43, "v_and sdwa WORD_1 PRESERVE", "v_and_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2") \
  X(44, "v_mov sdwa BYTE_1 PRESERVE", "v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2") \
  X(45, "v_and sdwa DWORD src BYTE", "v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2") \
  X(46, "v_msad_u8", "v_msad_u8 %0, %0, %1, %2")                                                       \
  X(47, "v_bfi_b32", "v_bfi_b32 %0, %1, %0, %2")                                                       \
  X(48, "v_perm_b32 (0, v, v)", "v_perm_b32 %0, 0, %1, %0")

template <int OP, bool CHAIN>
__global__ __launch_bounds__(1024) void k(unsigned *out, unsigned seed) {
  unsigned a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i + seed;
  unsigned b = seed | 1u, c = seed + 77u;
  asm volatile("s_mov_b32 s4, 3\n s_mov_b32 s5, 0x3f80" ::: "s4", "s5");
  for (int it = 0; it < 2000; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned &x = CHAIN ? a[r & 7] : a[i];   // CHAIN: 8 dependent instructions in a row on one register
#define X(N, NAME, ASM) \
        if (OP == N) asm volatile(ASM : "+v"(x) : "v"(b), "v"(c) : "vcc");
        OPS(X)
#undef X
      }
    }
  }
  unsigned r = 0;
  for (int i = 0; i < 8; ++i) r ^= a[i];
  if (r == 0x12345678u) out[0] = r;
}
template <int OP, bool CHAIN>
static double run(unsigned *d) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, CHAIN>), dim3(256), dim3(1024), 0, 0, d, 3u);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<OP, CHAIN>), dim3(256), dim3(1024), 0, 0, d, 5u);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  unsigned *d;
  (void)hipMalloc((void **)&d, 4);
  const double b0 = run<0, false>(d), b1 = run<0, true>(d);
  printf("%-28s %8s %8s   (x v_add_u32; 256k instructions per wave, 4 waves per SIMD)\n", "instruction", "indep", "chain");
#define X(N, NAME, ASM) \
  { const double t0 = run<N, false>(d), t1 = run<N, true>(d); printf("%-28s %8.2f %8.2f   %.3f ms\n", NAME, t0 / b0, t1 / b1, t0); }
  OPS(X)
#undef X
  return 0;
}
