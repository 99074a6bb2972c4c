/* quack — drop-in CLI entry point (quack.c:858). */
#include "quack_host.h"

int main(int argc, char **argv) { return qkh_main(argc, argv); }
