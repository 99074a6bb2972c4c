/*
 * dropin.c — libquack_dropin.so: the reference's seam (include/quack_dropin.h).
 *
 * Defines exactly the two functions quack.c's main() calls for the
 * accumulation path — read_adapters (quack.c:154-178) and read_fastq
 * (quack.c:180-228) — with the reference's argument and result types, on top of
 * the host pipeline (tokenizer -> pinned batches -> HIP kernels through the
 * C-ABI).  Everything else of libquack_host.so stays behind its own qkh_ names.
 */
#include <stdlib.h>
#include <string.h>

#include "quack_dropin.h"
#include "quack_host.h"

_Static_assert(sizeof(base_information) == sizeof(qk_base_info) && sizeof(base_information) == 776,
               "base_information must keep the reference's 97 x u64 image (quack.c:134-139)");

#define KMER_TABLE_SIZE (1u << 20)   /* pow(4, kmer_size), quack.c:155-156 */

static void die(const char *what, const char *path, const char *why) {
  fprintf(stderr, "quack: %s %s%s%s\n", what, path ? path : "(null)", why && *why ? ": " : "", why ? why : "");
  exit(1);
}

int *read_adapters(char *adapters_file) {
  uint32_t *bits = malloc(QK_KMER_TABLE_WORDS * sizeof *bits);
  int *kmers = malloc(KMER_TABLE_SIZE * sizeof *kmers);          /* quack.c:162 */
  if (!bits || !kmers) die("out of memory reading", adapters_file, "");
  if (!adapters_file || qkh_read_adapters(adapters_file, bits)) die("cannot read adapters file", adapters_file, "");
  for (uint32_t i = 0; i < KMER_TABLE_SIZE; i++) kmers[i] = (int)((bits[i >> 5] >> (i & 31)) & 1u);   /* quack.c:171 */
  free(bits);
  return kmers;
}

sequence_data *read_fastq(char *fastq_file, int *kmers) {
  sequence_data *out = malloc(sizeof *out);                      /* quack.c:190 */
  uint32_t *bits = NULL;
  qk_base_info *bases = NULL;
  uint64_t max_length = 0, number_of_sequences = 0;
  int devs[64], n_devs;
  if (!out) die("out of memory reading", fastq_file, "");
  if (kmers) {                                                   /* int kmers[4^10] -> the C-ABI's bitset */
    bits = calloc(QK_KMER_TABLE_WORDS, sizeof *bits);
    if (!bits) die("out of memory reading", fastq_file, "");
    for (uint32_t i = 0; i < KMER_TABLE_SIZE; i++)
      if (kmers[i]) bits[i >> 5] |= 1u << (i & 31);
  }
  n_devs = qkh_device_list(devs, 64);
  if (!fastq_file || qkh_accumulate_file(fastq_file, bits, devs, n_devs, &bases, &max_length, &number_of_sequences))
    die("cannot accumulate", fastq_file, qkh_last_error());
  free(bits);
  out->bases = (base_information *)bases;                        /* quack.c:224-226 */
  out->max_length = max_length;
  out->original_max_length = 0;                                  /* transform() sets it, quack.c:232 */
  out->number_of_sequences = number_of_sequences;
  return out;
}
