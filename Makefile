# Builds libpsn.so (HIP, gfx950) in-tree and the CPU oracle used by tests.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function
CSRC := poseidon_amd/csrc
LIB := poseidon_amd/libpsn.so

all: $(LIB) oracle

$(LIB): $(CSRC)/psn_api.hip $(CSRC)/psn_kernels.hip $(CSRC)/psn_device.h include/psn.h $(wildcard $(CSRC)/*.cpp) $(wildcard $(CSRC)/*.h)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/psn_api.hip $(wildcard $(CSRC)/*.cpp)

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
