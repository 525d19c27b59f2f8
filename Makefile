# Top-level build: libmcx.so (HIP, gfx950), the CPU oracle (test infrastructure) and the drivers.
HIPCC    ?= hipcc
ARCH     ?= gfx950
CSRC      = mcpar_amd/csrc
# -ffp-contract=off, no fast-math: "MCX arithmetic v1" is bit-reproducible only with explicit fma
HIPFLAGS  = -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -fno-fast-math \
            -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function -Iinclude

all: lib oracle

lib: mcpar_amd/libmcx.so

mcpar_amd/libmcx.so: $(CSRC)/mcx_engine.hip $(CSRC)/mcx_device.hpp $(CSRC)/mcx_numerics.hpp include/mcx.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/mcx_engine.hip

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f mcpar_amd/libmcx.so
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean

drivers: lib
	$(MAKE) -C mcpar_amd/drivers

test-cpu: lib oracle
	python -m pytest tests -x -q -m "not gpu"

.PHONY: drivers test-cpu
