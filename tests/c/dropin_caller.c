/*
 * dropin_caller.c — TEST PROGRAM: a stand-in for the reference's main()
 * (quack.c:858-928) that uses the accumulation path ONLY through the
 * reference's own seam: it re-declares base_information / sequence_data the way
 * quack.c does (quack.c:134-146), declares
 *     int* read_adapters(char*);  sequence_data* read_fastq(char*, int*);
 * as quack.c defines them (quack.c:154,180), and links libquack_dropin.so for
 * their definitions.  The counters it gets back go through the repo's
 * transform/draw restatement (qkh_render_document), so the output can be
 * compared byte for byte with the SVGs of the reference binary.
 *
 *   dropin_caller (-u reads | -1 fwd -2 rev) [-a adapters] [-n name]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- the reference's declarations, restated by the caller (NOT our header) ---- */
typedef struct {
  uint64_t scores[91];
  uint64_t content[4];
  uint64_t length_count;
  uint64_t kmer_count;
} base_information;

typedef struct {
  base_information *bases;
  uint64_t max_length;
  uint64_t original_max_length;
  uint64_t number_of_sequences;
} sequence_data;

int *read_adapters(char *adapters_file);
sequence_data *read_fastq(char *fastq_file, int *kmers);

/* ---- the repo's transform + draw (libquack_host.so) ---- */
struct qk_base_info_;
int qkh_render_document(FILE *out, FILE *err, const char *name, int adapters, void *fwd, uint64_t fwd_max_len,
                        uint64_t fwd_reads, void *rev, uint64_t rev_max_len, uint64_t rev_reads);

int main(int argc, char **argv) {
  char *fwd = NULL, *rev = NULL, *unp = NULL, *ad = NULL, *name = NULL;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "-1")) fwd = argv[i + 1];
    else if (!strcmp(argv[i], "-2")) rev = argv[i + 1];
    else if (!strcmp(argv[i], "-u")) unp = argv[i + 1];
    else if (!strcmp(argv[i], "-a")) ad = argv[i + 1];
    else if (!strcmp(argv[i], "-n")) name = argv[i + 1];
    else return 2;
  }
  const int paired = fwd && rev;
  if (paired == (unp != NULL)) return 2;
  int *kmers = ad ? read_adapters(ad) : NULL;                       /* quack.c:877 */
  sequence_data *a = read_fastq(paired ? fwd : unp, kmers);         /* quack.c:911 */
  sequence_data *b = paired ? read_fastq(rev, kmers) : NULL;        /* quack.c:917 */
  if (kmers) {                                                      /* the table really is int[4^10] of 0/1 */
    long set = 0;
    for (long i = 0; i < (1L << 20); i++) {
      if (kmers[i] != 0 && kmers[i] != 1) return 3;
      set += kmers[i];
    }
    if (getenv("DROPIN_VERBOSE")) fprintf(stderr, "kmers set: %ld\n", set);
  }
  if (getenv("DROPIN_VERBOSE"))
    fprintf(stderr, "read_fastq: max_length %llu number_of_sequences %llu\n", (unsigned long long)a->max_length,
            (unsigned long long)a->number_of_sequences);
  int rc = qkh_render_document(stdout, stderr, name, ad != NULL, a->bases, a->max_length, a->number_of_sequences,
                               b ? b->bases : NULL, b ? b->max_length : 0, b ? b->number_of_sequences : 0);
  free(a->bases);
  free(a);                                                          /* quack.c:914 */
  if (b) {
    free(b->bases);
    free(b);
  }
  free(kmers);
  return rc ? 1 : 0;
}
