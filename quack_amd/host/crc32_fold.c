/*
 * crc32_fold.c — gzip's CRC-32 (reflected polynomial 0xEDB88320) by carry-less
 * multiplication: 64 bytes per iteration are folded onto four 128-bit
 * accumulators (Gopal et al., "Fast CRC Computation for Generic Polynomials
 * Using PCLMULQDQ Instruction", Intel 2009), then 4 -> 1, 128 -> 64 bits and a
 * Barrett reduction.  ~15x zlib 1.2.11's table-driven crc32() on the hosts this
 * runs on, which matters because the inflate workers now check every member
 * (source.c): at 1 GB/s the check cost a third of their time.
 *
 * Used when the CPU has PCLMULQDQ + SSE4.1 (checked once); otherwise, and for
 * the unaligned head / short tail of a buffer, zlib's crc32().  Fuzzed against
 * zlib in tests/test_inflate.py.
 */
#include <stddef.h>
#include <stdatomic.h>
#include <stdint.h>
#include <zlib.h>

#include "inflate_fast.h"

#if defined(__x86_64__)
#include <immintrin.h>

__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_fold_blocks(uint32_t crc, const uint8_t *buf, size_t len) {
  /* len >= 64 and a multiple of 16; crc is the running value in zlib's convention already inverted by the caller */
  /* x^(512+64) mod P, x^512 mod P | x^(128+64) mod P, x^128 mod P | x^64 mod P | P', mu  (bit-reflected) */
  const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
  const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
  const __m128i k5k0 = _mm_set_epi64x(0x0000000000ll, 0x0163cd6124ll);
  const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;

  x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
  x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
  x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
  x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  buf += 64;
  len -= 64;
  while (len >= 64) {   /* fold the four accumulators over the next 64 bytes */
    x5 = _mm_clmulepi64_si128(x1, k1k2, 0x00);
    x6 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
    x7 = _mm_clmulepi64_si128(x3, k1k2, 0x00);
    x8 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
    x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
    x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
    x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
    y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
    x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
    x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
    buf += 64;
    len -= 64;
  }
  /* four accumulators -> one */
  x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00);
  x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00);
  x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00);
  x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (len >= 16) {   /* whole 16-byte blocks that are left */
    x2 = _mm_loadu_si128((const __m128i *)buf);
    x5 = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_clmulepi64_si128(x1, k3k4, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    buf += 16;
    len -= 16;
  }
  /* 128 -> 64 bits */
  x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8);
  x1 = _mm_xor_si128(x1, x2);
  x0 = k5k0;
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, x3);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  /* Barrett reduction 64 -> 32 bits */
  x0 = poly;
  x2 = _mm_and_si128(x1, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}

static int have_clmul(void) {
  /* (asked by every producer thread: an atomic, although all of them would store the same value) */
  static _Atomic int known = -1;
  int k = atomic_load_explicit(&known, memory_order_relaxed);
  if (k < 0) {
    k = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    atomic_store_explicit(&known, k, memory_order_relaxed);
  }
  return k;
}
#endif

/* crc32() of zlib, same convention: crc = qkh_crc32(crc, buf, len), start with 0 */
uint32_t qkh_crc32(uint32_t crc, const uint8_t *buf, size_t len) {
#if defined(__x86_64__)
  if (len >= 128 && have_clmul()) {
    const size_t body = len & ~(size_t)15;
    crc = ~crc32_fold_blocks(~crc, buf, body);
    buf += body;
    len -= body;
  }
#endif
  while (len) {   /* (zlib's length is 32 bits) */
    const size_t step = len > ((size_t)1 << 30) ? ((size_t)1 << 30) : len;
    crc = (uint32_t)crc32(crc, buf, (uInt)step);
    buf += step;
    len -= step;
  }
  return crc;
}
