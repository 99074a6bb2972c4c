// tools/isa_probe.hip — what a few gfx950 instructions compute, checked on the device before a kernel relies on them
// (round 5: v_msad_u8's masking operand, v_perm_b32 as an 8-entry byte table, SDWA byte insertion with UNUSED_PRESERVE).
//   hipcc --offload-arch=gfx950 -O2 -o tools/isa_probe tools/isa_probe.hip && tools/isa_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(uint32_t *out) {
  const uint32_t t = threadIdx.x;
  // msad: which operand masks?  a = 0x05000700 (bytes 0,7,0,5), b = 0x00030003 (3,0,3,0)
  const uint32_t a = 0x05000700u + t * 0u, b = 0x00030003u + t * 0u;
  out[0] = __builtin_amdgcn_msad_u8(a, b, 100u);   // mask on src1 (b != 0): |0-3| + |0-3| = 6 -> 106; on src0: |7-0| + |5-0| = 12 -> 112
  out[1] = __builtin_amdgcn_msad_u8(b, a, 100u);
  out[2] = __builtin_amdgcn_sad_u8(a, b, 100u);     // 3 + 7 + 3 + 5 = 18 -> 118
  // perm as a table: selector byte i picks byte sel[i] of {src0 (bytes 4-7), src1 (bytes 0-3)}
  const uint32_t hi = 0x77665544u, lo = 0x33221100u;
  out[3] = __builtin_amdgcn_perm(hi, lo, 0x07040300u + t * 0u);   // -> 0x77 0x44 0x33 0x00 = 0x77443300
  out[4] = __builtin_amdgcn_perm(hi, lo, 0x0C0D0501u + t * 0u);   // 0x0C -> 0x00, 0x0D -> 0xFF: 0x00FF5511
  // SDWA: byte 2 of src into byte 1 of dst, the rest of dst preserved
  uint32_t d = 0xAABBCCDDu + t * 0u;
  const uint32_t src = 0x11223344u + t * 0u;
  asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(d) : "v"(src));
  out[5] = d;   // 0xAABB22DD
  // udot4 with byte weights 64,16,4,1
  out[6] = __builtin_amdgcn_udot4(0x03020100u + t * 0u, 0x01041040u, 0u, false);   // 0*64 + 1*16 + 2*4 + 3*1 = 27
}

int main() {
  uint32_t *d = nullptr, h[8] = {};
  if (hipMalloc((void **)&d, sizeof h) != hipSuccess) return 2;
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  const char *names[] = {"msad(a,b,100) [106: src1 masks, 112: src0 masks]", "msad(b,a,100)", "sad(a,b,100) [118]", "perm table [77443300]",
                         "perm consts [00ff5511]", "sdwa byte insert [aabb22dd]", "udot4 weights [27]"};
  for (int i = 0; i < 7; ++i) printf("%-52s = %u (0x%08x)\n", names[i], h[i], h[i]);
  return 0;
}
