/*
 * cli.c — quack's command line, restated (parse_options quack.c:59-132,
 * main quack.c:858-928).  Same argv grammar, same stdout / stderr bytes, same
 * exit codes.  Device selection is environment-only so the CLI stays a
 * drop-in:  QUACK_DEVICES=0,1,...  (default: device 0).
 *
 * One deliberate difference: the reference prints the <svg> envelope before it
 * reads the FASTQ and dies with SIGSEGV on an unreadable file (stdout empty,
 * rc 139).  Here every input is accumulated first; on any failure a message
 * goes to stderr, nothing to stdout, and the exit code is 1.
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "quack_host.h"

/* one read_fastq() (quack.c:911 / 917); the two mates of a pair are independent
 * accumulations, so they run concurrently — printed forward first regardless */
typedef struct {
  const char *path;
  const uint32_t *bitset;
  const int *devs;
  int n_devs;
  qk_base_info *tab;
  uint64_t max_len, n_reads;
  int rc;
  char err[512];
} file_job;

static void *file_job_main(void *arg) {
  file_job *j = arg;
  j->rc = qkh_accumulate_file(j->path, j->bitset, j->devs, j->n_devs, &j->tab, &j->max_len, &j->n_reads);
  if (j->rc) snprintf(j->err, sizeof j->err, "%s", qkh_last_error());
  return NULL;
}

/* the one or two read_fastq() calls of a run */
static void run_jobs(file_job *jobs, int n_jobs) {
  pthread_t th;
  int threaded = 0;
  if (n_jobs == 2 && pthread_create(&th, NULL, file_job_main, &jobs[1]) == 0) threaded = 1;
  file_job_main(&jobs[0]);
  if (n_jobs == 2) {
    if (threaded) pthread_join(th, NULL);
    else file_job_main(&jobs[1]);
  }
}

static int write_all(int fd, const void *buf, size_t n) {
  const char *p = buf;
  while (n) {
    ssize_t k = write(fd, p, n);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return -1;
    p += k;
    n -= (size_t)k;
  }
  return 0;
}
static int read_all(int fd, void *buf, size_t n) {
  char *p = buf;
  while (n) {
    ssize_t k = read(fd, p, n);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return -1;
    p += k;
    n -= (size_t)k;
  }
  return 0;
}

/* The accumulation in a WORKER PROCESS (round 3).  A process that has used the GPU takes 0.13-0.15 s to exit —
 * the kernel driver tearing down its queues, pinned and device memory — a quarter of a run on a 0.6-Gbase file, and
 * nothing the user waits for serves a purpose: the counters are complete.  So everything that touches HIP happens in
 * a child forked before the runtime starts; it sends the tables back through a pipe (116 KB at 150 bp), closes its
 * standard streams — so that nobody reading them waits for it — and exits on its own time, while this process, which
 * never loaded the runtime, transforms, draws and returns.  QUACK_NO_FORK=1 (or QUACK_FULL_TEARDOWN=1, the leak
 * checkers' mode) keeps everything in one process.  Returns 0 when the jobs' results are in `jobs`, -1 when there is
 * no worker (the caller then runs them itself). */
static int run_jobs_in_worker(file_job *jobs, int n_jobs) {
  int fd[2];
  pid_t pid;
  /* only in a process that said it is quack's own and about to exit (main.c): a caller that runs qkh_main inside a larger
   * program may have the HIP runtime up already, and a process must not fork behind it */
  if (!qkh_process_exits() || getenv("QUACK_NO_FORK") || getenv("QUACK_FULL_TEARDOWN")) return -1;
  if (pipe(fd)) return -1;
  fflush(stdout);
  fflush(stderr);
  pid = fork();
  if (pid < 0) {
    close(fd[0]);
    close(fd[1]);
    return -1;
  }
  if (pid == 0) {
    int ok = 1;
    close(fd[0]);
    (void)dup2(2, 1);   /* the document is the parent's to write: whatever a library prints to stdout here goes to stderr */
    run_jobs(jobs, n_jobs);
    if (getenv("QUACK_TEST_WORKER_DIES")) abort();   /* (tests: the parent's report of a worker that is gone) */
    for (int k = 0; k < n_jobs && ok; k++) {
      ok = !write_all(fd[1], &jobs[k].rc, sizeof jobs[k].rc) && !write_all(fd[1], jobs[k].err, sizeof jobs[k].err) &&
           !write_all(fd[1], &jobs[k].max_len, sizeof jobs[k].max_len) && !write_all(fd[1], &jobs[k].n_reads, sizeof jobs[k].n_reads);
      if (ok && !jobs[k].rc && jobs[k].max_len)
        ok = !write_all(fd[1], jobs[k].tab, (size_t)jobs[k].max_len * sizeof(qk_base_info));
    }
    fflush(stderr);
    close(fd[1]);
    /* nobody waits for this process any more: let go of the standard streams before the slow part of exiting */
    close(0);
    close(1);
    close(2);
    _exit(ok ? 0 : 3);
  }
  close(fd[1]);
  for (int k = 0; k < n_jobs; k++) {
    jobs[k].tab = NULL;
    if (read_all(fd[0], &jobs[k].rc, sizeof jobs[k].rc) || read_all(fd[0], jobs[k].err, sizeof jobs[k].err) ||
        read_all(fd[0], &jobs[k].max_len, sizeof jobs[k].max_len) || read_all(fd[0], &jobs[k].n_reads, sizeof jobs[k].n_reads)) {
      /* the worker is gone without a word: say how it ended */
      int st = 0;
      (void)waitpid(pid, &st, 0);
      jobs[k].rc = -1;
      jobs[k].max_len = 0;
      if (WIFSIGNALED(st)) snprintf(jobs[k].err, sizeof jobs[k].err, "%s: the accumulation process was killed by signal %d", jobs[k].path, WTERMSIG(st));
      else snprintf(jobs[k].err, sizeof jobs[k].err, "%s: the accumulation process ended without a result (status %d)", jobs[k].path, WEXITSTATUS(st));
      for (int j = k + 1; j < n_jobs; j++) {
        jobs[j].rc = -1;
        jobs[j].max_len = 0;
        snprintf(jobs[j].err, sizeof jobs[j].err, "%s: not read", jobs[j].path);
      }
      break;
    }
    jobs[k].err[sizeof jobs[k].err - 1] = 0;
    if (!jobs[k].rc && jobs[k].max_len) {
      jobs[k].tab = calloc(jobs[k].max_len, sizeof(qk_base_info));
      if (!jobs[k].tab || read_all(fd[0], jobs[k].tab, (size_t)jobs[k].max_len * sizeof(qk_base_info))) {
        free(jobs[k].tab);
        jobs[k].tab = NULL;
        jobs[k].rc = -1;
        snprintf(jobs[k].err, sizeof jobs[k].err, "%s: the counters did not arrive", jobs[k].path);
      }
    }
  }
  close(fd[0]);
  return 0;
}

static const char *const k_version = "quack 1.1.1";   /* quack.c:54 */

static const char *const k_help =                     /* quack.c:70-80 */
    "Usage: quack [OPTION...]\n"
    "quack -- A FASTQ quality assessment tool\n\n"
    "  -1, --forward file.1.fq.gz      Forward strand\n"
    "  -2, --reverse file.2.fq.gz      Reverse strand\n"
    "  -a, --adapters adapters.fa.gz   (Optional) Adapters file\n"
    "  -n, --name NAME                 (Optional) Display in output\n"
    "  -u, --unpaired unpaired.fq.gz   Data (only use with -u)\n"
    "  -?, --help                      Give this help list\n"
    "      --usage                     (use alone)\n"
    "  -V, --version                   Print program version (use alone)\n"
    "Report bugs to <thrash@igbb.msstate.edu>.\n";

static const char *const k_bad_option =               /* quack.c:113-123 */
    "Usage: quack [OPTION...]\n"
    "quack -- A FASTQ quality assessment tool\n\n"
    "  -1, --forward file.1.fq.gz      Forward strand\n"
    "  -2, --reverse file.2.fq.gz      Reverse strand\n"
    "  -a, --adapters adapters.fa.gz    Adapters file\n"
    "  -n, --name NAME            Display in output\n"
    "  -u, --unpaired unpaired.fq.gz        Data (only use with -u)\n"
    "  -?, --help                 Give this help list\n"
    "      --usage                (use alone)\n"
    "  -V, --version              Print program version (use alone)\n"
    "Report bugs to <thrash@igbb.msstate.edu>.\n";

static const char *const k_try_help =                 /* quack.c:873 */
    "Usage: quack [OPTION...]\nTry `quack --help' or `quack --usage' for more information.";

typedef struct {
  const char *name, *forward, *reverse, *unpaired, *adapters;
} options;

static int is_opt(const char *arg, const char *s, const char *l) {
  return strcmp(arg, s) == 0 || strcmp(arg, l) == 0;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int qkh_main(int argc, char **argv) {
  const double t_main = now_s();
  double t_jobs = 0;
  options o = {0};
  uint32_t *bitset = NULL;
  qk_base_info *tab[2] = {NULL, NULL};
  uint64_t max_len[2] = {0, 0}, n_reads[2] = {0, 0};
  int devs[64], n_devs, paired, unpaired, rc = 1;

  if (argc == 1 || argc == 2) {                       /* quack.c:68-87 */
    if (argc == 1 || is_opt(argv[1], "-?", "--help") || strcmp(argv[1], "--usage") == 0)
      fputs(k_help, stdout);
    if (argc == 2 && is_opt(argv[1], "-V", "--version")) {
      printf("%s\n", k_version);
      return 0;
    }
  }
  if (argc > 2 && argc % 2 != 0) {                    /* quack.c:89-130 */
    for (int i = 1; i < argc; i += 2) {
      const char *flag = argv[i], *val = argv[i + 1];
      if (is_opt(flag, "-1", "--forward")) o.forward = val;
      else if (is_opt(flag, "-2", "--reverse")) o.reverse = val;
      else if (is_opt(flag, "-a", "--adapters")) o.adapters = val;
      else if (is_opt(flag, "-u", "--unpaired")) o.unpaired = val;
      else if (is_opt(flag, "-n", "--name")) o.name = val;
      else {
        fputs(k_bad_option, stderr);
        return EXIT_FAILURE;
      }
    }
  }
  paired = o.forward != NULL && o.reverse != NULL;    /* quack.c:867-875 */
  unpaired = o.unpaired != NULL;
  if (paired == unpaired) {
    printf("%s\n", k_try_help);
    return 1;
  }

  if (o.adapters) {                                   /* quack.c:877 */
    bitset = malloc(QK_KMER_TABLE_WORDS * sizeof(uint32_t));
    if (!bitset || qkh_read_adapters(o.adapters, bitset)) {
      fprintf(stderr, "quack: cannot read adapters file %s\n", o.adapters);
      goto done;
    }
  }
  n_devs = qkh_device_list(devs, 64);
  if (getenv("QUACK_INIT_FIRST")) {   /* experiment: the HIP runtime comes up before the decoder threads exist */
    int n = 0;
    const double t = now_s();
    (void)qk_device_count(&n);
    if (getenv("QUACK_VERBOSE")) fprintf(stderr, "[quack] main: HIP runtime up in %.3f s (alone)\n", now_s() - t);
  }

  /* accumulate first (quack.c:911,917), print afterwards */
  {
    file_job jobs[2];
    int n_jobs = paired ? 2 : 1;
    memset(jobs, 0, sizeof jobs);
    jobs[0].path = paired ? o.forward : o.unpaired;
    jobs[1].path = o.reverse;
    for (int k = 0; k < n_jobs; k++) {
      jobs[k].bitset = bitset;
      jobs[k].devs = devs;
      jobs[k].n_devs = n_devs;
    }
    if (run_jobs_in_worker(jobs, n_jobs)) run_jobs(jobs, n_jobs);
    for (int k = 0; k < n_jobs; k++) {
      tab[k] = jobs[k].tab;
      max_len[k] = jobs[k].max_len;
      n_reads[k] = jobs[k].n_reads;
    }
    for (int k = 0; k < n_jobs; k++) {
      if (jobs[k].rc) {
        fprintf(stderr, "quack: %s\n", jobs[k].err);
        goto done;
      }
      if (max_len[k] == 0) {
        fprintf(stderr, "quack: %s: no sequence data\n", jobs[k].path);
        goto done;
      }
    }
  }
  t_jobs = now_s();
  if (qkh_render_document(stdout, stderr, o.name, o.adapters != NULL, tab[0], max_len[0], n_reads[0],
                          paired ? tab[1] : NULL, max_len[1], n_reads[1])) {
    fprintf(stderr, "quack: nothing to draw\n");
    goto done;
  }
  rc = 0;                                             /* quack.c:927 */
  if (getenv("QUACK_VERBOSE"))
    fprintf(stderr, "[quack] main: accumulate %.3f s, transform+draw %.3f s\n", t_jobs - t_main, now_s() - t_jobs);
done:
  free(tab[0]);
  free(tab[1]);
  free(bitset);
  return rc;
}
