# Builds the gfx950 shared library and the C oracle (test infrastructure).
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := lshm_amd/csrc
OBJDIR := build/obj
LIB := lshm_amd/lib/liblshm_hip.so
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HIPFLAGS := -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Wall -Wno-unused-function

all: $(LIB) oracle testlibs

$(OBJDIR)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/kernels.h include/lshm.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# headers of single kernel families (kept out of kernels.h, which rebuilds everything)
$(OBJDIR)/deep2d.o $(OBJDIR)/capi.o $(OBJDIR)/engine.o: $(CSRC)/deep2d.h
$(OBJDIR)/chain1d_full.o $(OBJDIR)/capi.o $(OBJDIR)/engine.o: $(CSRC)/chain1d_full.h
$(OBJDIR)/chain1d.o $(OBJDIR)/chain1d_full.o: $(CSRC)/chain1d_dev.h

$(LIB): $(OBJS)
	@mkdir -p lshm_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle: oracle/_build/liblshm_oracle_c.so

oracle/_build/liblshm_oracle_c.so: oracle/lshm_oracle_c.c
	@mkdir -p oracle/_build
	gcc -O2 -fPIC -shared -fopenmp -o $@ $< -lm

# the C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: the pool has no GPU sanitizer, the HIP library no CPU build)
sanitize: oracle/_build/oracle_asan
	ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 ./oracle/_build/oracle_asan

oracle/_build/oracle_asan: oracle/sanitize_main.c oracle/lshm_oracle_c.c
	@mkdir -p oracle/_build
	gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Wall -Wno-unknown-pragmas -o $@ oracle/sanitize_main.c -lm

# test infrastructure: a host-shared-memory stand-in for RCCL (two ranks on one GPU, tests/test_gpu_dp.py)
testlibs: tests/fake_rccl/libfake_rccl.so

tests/fake_rccl/libfake_rccl.so: tests/fake_rccl/fake_rccl.cpp
	$(HIPCC) -O2 -fPIC -shared -std=c++17 --offload-arch=$(ARCH) -o $@ $< -lrt

clean:
	rm -rf build lshm_amd/lib oracle/_build tests/fake_rccl/libfake_rccl.so

.PHONY: all oracle testlibs clean sanitize
