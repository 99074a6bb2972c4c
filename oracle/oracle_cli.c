/*
 * oracle_cli.c — TEST INFRASTRUCTURE.  Dumps the oracle's raw counters as text:
 *   quack_oracle dump reads.fq[.gz] [adapters.fa[.gz]]
 *   quack_oracle time reads.fq[.gz] [adapters.fa[.gz]]   (seconds for read_fastq)
 * Output of `dump`: "nseq N maxlen L" then one line per non-zero counter:
 *   "<position> <row> <count>"  (row: 0-90 score, 91-94 A,T,C,G, 95 length, 96 kmer)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "quack_oracle.h"

int main(int argc, char **argv) {
  oracle_table t;
  uint8_t *kmers = NULL;
  if (argc < 3 || (strcmp(argv[1], "dump") && strcmp(argv[1], "time"))) {
    fprintf(stderr, "usage: %s dump|time reads.fq[.gz] [adapters.fa[.gz]]\n", argv[0]);
    return 2;
  }
  if (argc > 3) {
    kmers = malloc(ORACLE_KMER_TABLE);
    if (!kmers || oracle_read_adapters(argv[3], kmers)) {
      fprintf(stderr, "cannot read adapters %s\n", argv[3]);
      return 1;
    }
  }
  oracle_table_init(&t);
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  if (oracle_read_fastq(argv[2], kmers, &t)) {
    fprintf(stderr, "cannot read %s\n", argv[2]);
    return 1;
  }
  clock_gettime(CLOCK_MONOTONIC, &b);
  if (!strcmp(argv[1], "time")) {
    printf("%.6f\n", (b.tv_sec - a.tv_sec) + (b.tv_nsec - a.tv_nsec) * 1e-9);
  } else {
    printf("nseq %llu maxlen %llu\n", (unsigned long long)t.number_of_sequences,
           (unsigned long long)t.max_length);
    for (uint64_t p = 0; p < t.max_length; p++)
      for (int r = 0; r < ORACLE_ROWS; r++)
        if (t.bases[p * ORACLE_ROWS + r])
          printf("%llu %d %llu\n", (unsigned long long)p, r,
                 (unsigned long long)t.bases[p * ORACLE_ROWS + r]);
  }
  oracle_table_free(&t);
  free(kmers);
  return 0;
}
