/* quack — drop-in CLI entry point (quack.c:858). */
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include "quack_host.h"

int main(int argc, char **argv) {
  /* tearing the HIP runtime down (pinned slots, device buffers, the runtime's own exit handlers) takes
   * ~0.16 s and serves nothing in a process that is done: print, flush, _exit.
   * QUACK_FULL_TEARDOWN=1 keeps the orderly path (leak checkers in the tests). */
  const int fast = getenv("QUACK_FULL_TEARDOWN") == NULL;
  int rc;
  qkh_process_exits_after_this(fast);
  rc = qkh_main(argc, argv);
  if (!fast) return rc;
  fflush(stdout);
  fflush(stderr);
  _exit(rc);
}
