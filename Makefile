# quack-mi355x — top-level build (gfx950 only).
#   make            libquack_hip.so (HIP kernels + C-ABI), host library + CLI
#   make oracle     test oracle (plain C restatement, CPU)
#   make tools      synthetic FASTQ generator, kbench
HIPCC ?= /opt/rocm/bin/hipcc
CC ?= gcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Iinclude -Wall -Wno-unused-function
CFLAGS ?= -O2 -g -fPIC -std=c11 -Wall -Wextra -Iinclude -D_GNU_SOURCE

CSRC := quack_amd/csrc
HOST := quack_amd/host
LIB_HIP := quack_amd/libquack_hip.so
# the experiment build: the same source with -DQK_EXPERIMENT — the QUACK_HIP_TUNE switches and the kernel variants of launch
# geometries the planner does not pick (tools, and the parity tests that cross-check those geometries); never linked by the host
LIB_HIP_EXP := quack_amd/libquack_hip_exp.so
KERNEL_HDRS := $(CSRC)/qk_kernels.hip.h $(CSRC)/qk_adapter_kernels.hip.h

.PHONY: all hip host exp oracle tools clean
all: hip host exp

hip: $(LIB_HIP)
$(LIB_HIP): $(CSRC)/qk_shim.hip $(KERNEL_HDRS) include/quack_hip.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/qk_shim.hip -ldl

exp: $(LIB_HIP_EXP)
$(LIB_HIP_EXP): $(CSRC)/qk_shim.hip $(KERNEL_HDRS) include/quack_hip.h
	$(HIPCC) $(HIPFLAGS) -DQK_EXPERIMENT -shared -o $@ $(CSRC)/qk_shim.hip -ldl

host: $(LIB_HIP)
	$(MAKE) -C $(HOST)

oracle:
	$(MAKE) -C oracle

tools: tools/kbench tools/gen_fastq tools/inflate_bench tools/feed_bench
tools/feed_bench: tools/feed_bench.c host
	$(CC) -O3 -o $@ tools/feed_bench.c -Iinclude -I$(HOST) -Lquack_amd -lquack_host -lquack_hip -Wl,-rpath,'$$ORIGIN/../quack_amd'
tools/inflate_bench: tools/inflate_bench.c $(HOST)/pinflate.c $(HOST)/inflate_fast.c $(HOST)/inflate_body.inc $(HOST)/inflate_fast.h $(HOST)/crc32_fold.c
	$(CC) -O3 -o $@ tools/inflate_bench.c $(HOST)/pinflate.c $(HOST)/inflate_fast.c $(HOST)/crc32_fold.c -I$(HOST) -lz -lpthread
tools/gen_fastq: tools/gen_fastq.c
	$(CC) -O2 -o $@ $< -lz
tools/kbench: tools/kbench.cpp $(CSRC)/qk_shim.hip $(KERNEL_HDRS) include/quack_hip.h
	$(HIPCC) $(HIPFLAGS) -DQK_ABLATION -DQK_EXPERIMENT -o $@ tools/kbench.cpp $(CSRC)/qk_shim.hip -ldl

clean:
	rm -f $(LIB_HIP) $(LIB_HIP_EXP) tools/kbench tools/gen_fastq tools/inflate_bench tools/feed_bench
	-$(MAKE) -C $(HOST) clean
	-$(MAKE) -C oracle clean
