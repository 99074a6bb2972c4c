/* oracle/ref_peek.c — TEST INFRASTRUCTURE (build container only).
 *
 * An LD_PRELOAD observer for the UNMODIFIED reference binary
 * (/root/reference/bin/Linux_x86_64_kernel_3.10.0/quack): it makes the
 * reference hand out its RAW counter table — `bases[]` as read_fastq leaves it
 * (quack.c:223-226), before transform (quack.c:230) turns counts into
 * percentages — so that the oracle and the HIP path can be compared with the
 * reference's integers directly, not only with what survives into the SVG.
 *
 * How: read_fastq grows `bases` with realloc(bases, l * sizeof(base_information))
 * (quack.c:195) and sizeof(base_information) == 776 == 8 * 97, a factor no
 * other allocation of the program has (kseq's buffers are powers of two, the
 * k-mer table is 4 MiB).  The observer remembers the last realloc whose size
 * is a positive multiple of 776 and, when the file is closed (gzclose,
 * quack.c:223 — the next statement after the loop), writes that block to
 *     $QUACK_PEEK_OUT.<k>        k = 0, 1: forward / reverse file
 * and forgets it.  read_adapters' gzclose (quack.c:176) finds no block and
 * writes nothing.  Nothing of the reference is modified, linked or copied;
 * nothing here ships (see oracle/make_raw_goldens.sh).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define QK_POSITION_BYTES 776u   /* sizeof(base_information), quack.c:134-139 */

static void *(*real_realloc)(void *, size_t);
static int (*real_gzclose)(void *);
static void *last_ptr;
static size_t last_size;
static int dumps;

void *realloc(void *p, size_t n) {
  if (!real_realloc) real_realloc = (void *(*)(void *, size_t))dlsym(RTLD_NEXT, "realloc");
  void *r = real_realloc(p, n);
  if (r && n && n % QK_POSITION_BYTES == 0) {
    last_ptr = r;
    last_size = n;
  } else if (p && p == last_ptr) {
    last_ptr = NULL;   /* the block was resized to something else: not the table */
    last_size = 0;
  }
  return r;
}

int gzclose(void *f) {
  if (!real_gzclose) real_gzclose = (int (*)(void *))dlsym(RTLD_NEXT, "gzclose");
  const char *out = getenv("QUACK_PEEK_OUT");
  if (out && last_ptr) {
    char path[4096];
    snprintf(path, sizeof path, "%s.%d", out, dumps++);
    FILE *fp = fopen(path, "wb");
    if (!fp || fwrite(last_ptr, 1, last_size, fp) != last_size || fclose(fp) != 0) {
      fprintf(stderr, "ref_peek: cannot write %s\n", path);
      _Exit(97);
    }
  }
  last_ptr = NULL;
  last_size = 0;
  return real_gzclose(f);
}
