/*
 * quack_dropin.h — the reference's own seam, as linkable symbols
 * ("libquack_dropin.so").
 *
 * IGBB/quack has no plugin API: the accumulation path sits behind two plain C
 * functions with external linkage and two structs (quack.c:134-146,154,180).
 * This header re-declares exactly those, and libquack_dropin.so defines exactly
 * those two functions on top of the MI355X path (include/quack_hip.h), so that
 * quack.c can drop its own definitions (quack.c:154-228) and link this library
 * instead; transform() and draw() (quack.c:230-856) run unmodified on the
 * result.  See INTEGRATION.md §2.
 *
 *     int*           read_adapters(char *adapters_file);         quack.c:154
 *     sequence_data* read_fastq(char *fastq_file, int *kmers);   quack.c:180
 *
 * Ownership is the reference's: read_adapters returns a malloc'ed int[4^10]
 * (quack.c:162, never freed by quack.c); read_fastq returns a malloc'ed
 * sequence_data whose `bases` is a malloc'ed array of max_length entries
 * (quack.c:190,195; main frees only the struct, quack.c:914,920).
 * original_max_length is left for transform() to fill (quack.c:232), as in the
 * reference.  Devices: QUACK_DEVICES=0,1,... (default 0), like the CLI.
 * Errors: the reference checks nothing and crashes on an unreadable file
 * (quack.c:160-161,187-188); these functions print a message to stderr and
 * exit(1) — there is no CPU fallback.
 */
#ifndef QUACK_DROPIN_H
#define QUACK_DROPIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {                 /* quack.c:134-139 */
  uint64_t scores[91];
  uint64_t content[4];
  uint64_t length_count;
  uint64_t kmer_count;
} base_information;

typedef struct {                 /* quack.c:141-146 */
  base_information *bases;
  uint64_t max_length;
  uint64_t original_max_length;
  uint64_t number_of_sequences;
} sequence_data;

int *read_adapters(char *adapters_file);
sequence_data *read_fastq(char *fastq_file, int *kmers);

#ifdef __cplusplus
}
#endif
#endif /* QUACK_DROPIN_H */
