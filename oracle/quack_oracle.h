/*
 * quack_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's accumulation path, used only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
 * Nothing under quack_amd/ may include, link or call this.
 *
 * Restates (file:line in /root/reference):
 *   lookup / base->bin            quack.c:148-150, 201
 *   read_adapters                 quack.c:154-178
 *   read_fastq                    quack.c:180-228
 *   base_information              quack.c:134-139
 * and the tokenizer the reference gets from klib kseq.h (KSEQ_INIT at
 * quack.c:152; attractivechaos/klib, un-vendored submodule, version unpinned —
 * its published kseq_read algorithm is restated in quack_oracle.c).
 *
 * Pinning: tests/test_oracle_pins.py recovers the raw integer counters from
 * SVGs produced by the reference's own prebuilt binary
 * (/root/reference/bin/Linux_x86_64_kernel_3.10.0/quack, run by
 * oracle/make_goldens.sh) and compares them with this oracle bit for bit, plus
 * the known-answer case of SURVEY.md §8c.
 */
#ifndef QUACK_ORACLE_H
#define QUACK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_ROWS 97         /* 91 scores + 4 content + length + kmer */
#define ORACLE_KMER_TABLE (1u << 20)

typedef struct {
  uint64_t *bases;             /* [max_length][ORACLE_ROWS], reference AoS order */
  uint64_t max_length;
  uint64_t number_of_sequences;
  uint64_t capacity;           /* allocated positions */
} oracle_table;

void oracle_table_init(oracle_table *t);
void oracle_table_free(oracle_table *t);

/* quack.c:148-150,201 on its defined domain; total elsewhere (see .c) */
int oracle_base_code(unsigned char c);
/* quack.c:203: byte-33; returns -1 where the reference indexes out of range */
int oracle_qual_bin(unsigned char b);

/* read_adapters (quack.c:154-178).  kmers: ORACLE_KMER_TABLE bytes (0/1). */
void oracle_adapter_insert(uint8_t *kmers, const char *seq, size_t len);
int oracle_read_adapters(const char *path, uint8_t *kmers);

/* One read of read_fastq's loop body (quack.c:194-220). kmers may be NULL. */
int oracle_accumulate_read(oracle_table *t, const uint8_t *seq,
                           const uint8_t *qual, size_t len,
                           const uint8_t *kmers);
/* A whole batch in the C-ABI's layout (offsets == NULL: fixed read_len). */
int oracle_accumulate_batch(oracle_table *t, const uint8_t *seq,
                            const uint8_t *qual, const uint64_t *offsets,
                            uint64_t n_reads, uint32_t read_len,
                            const uint8_t *kmers);
/* read_fastq (quack.c:180-228): tokenise a (gz) FASTQ/FASTA file. */
int oracle_read_fastq(const char *path, const uint8_t *kmers, oracle_table *t);

/* ---- tokenizer with kseq_read semantics (exposed for differential tests) */
typedef struct oracle_reader oracle_reader;
oracle_reader *oracle_reader_open(const char *path);
/* >=0: sequence length; -1 EOF; -2 truncated / length mismatch */
long oracle_reader_next(oracle_reader *r, const uint8_t **seq,
                        const uint8_t **qual);
void oracle_reader_close(oracle_reader *r);

#ifdef __cplusplus
}
#endif
#endif
