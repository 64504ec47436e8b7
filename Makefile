# Builds the gfx950 shared library and the C oracle (test infrastructure).
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := lshm_amd/csrc
OBJDIR := build/obj
LIB := lshm_amd/lib/liblshm_hip.so
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HIPFLAGS := -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Wall -Wno-unused-function

all: $(LIB) oracle

$(OBJDIR)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/kernels.h include/lshm.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p lshm_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle: oracle/_build/liblshm_oracle_c.so

oracle/_build/liblshm_oracle_c.so: oracle/lshm_oracle_c.c
	@mkdir -p oracle/_build
	gcc -O2 -fPIC -shared -fopenmp -o $@ $< -lm

clean:
	rm -rf build lshm_amd/lib oracle/_build

.PHONY: all oracle clean
