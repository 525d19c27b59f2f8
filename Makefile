# Top-level build: libmcx.so (HIP, gfx950), the CPU oracle (test infrastructure) and the drivers.
HIPCC    ?= hipcc
ARCH     ?= gfx950
CSRC      = mcpar_amd/csrc
# -ffp-contract=off, no fast-math: the MCX arithmetic is bit-reproducible only with explicit fma
HIPFLAGS  = -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -fno-fast-math \
            -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function -Iinclude

all: lib oracle

lib:
	$(MAKE) -j6 mcpar_amd/libmcx.so

OBJS = $(CSRC)/mcx_engine.o $(CSRC)/mcx_plan.o $(CSRC)/mcx_exchange.o $(CSRC)/mcx_sink.o $(CSRC)/mcx_murray.o $(CSRC)/mcx_k_fast.o $(CSRC)/mcx_k_fastb.o $(CSRC)/mcx_k_fastb_full.o $(CSRC)/mcx_k_fast_full.o $(CSRC)/mcx_k_pregen.o $(CSRC)/mcx_k_generic_burn.o \
       $(CSRC)/mcx_k_generic_main.o $(CSRC)/mcx_k_persist.o $(CSRC)/mcx_user.o
HDRS = $(CSRC)/mcx_device.hpp $(CSRC)/mcx_numerics.hpp $(CSRC)/mcx_launch.hpp $(CSRC)/mcx_persist.hpp $(CSRC)/mcx_engine_internal.hpp include/mcx.h

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

# the kernel headers as strings inside the library: a user's likelihood source is compiled against them at run time (hiprtc)
$(CSRC)/mcx_rtc_headers.inc: $(CSRC)/mcx_numerics.hpp $(CSRC)/mcx_device.hpp $(CSRC)/mcx_fastb.hpp $(CSRC)/mcx_persist.hpp tools/embed_headers.py
	python3 tools/embed_headers.py $@ k_hdr_mcx_numerics=$(CSRC)/mcx_numerics.hpp k_hdr_mcx_device=$(CSRC)/mcx_device.hpp k_hdr_mcx_fastb=$(CSRC)/mcx_fastb.hpp k_hdr_mcx_persist=$(CSRC)/mcx_persist.hpp
$(CSRC)/mcx_user.o: $(CSRC)/mcx_rtc_headers.inc

# what only one translation unit sees
$(CSRC)/mcx_murray.o: $(CSRC)/mcx_remote.hpp $(CSRC)/mcx_cull_proj.hpp $(CSRC)/mcx_screen.hpp
$(CSRC)/mcx_sink.o: $(CSRC)/mcx_text.hpp $(CSRC)/fmt_g6.hpp
$(CSRC)/mcx_k_fastb.o $(CSRC)/mcx_k_fastb_full.o: $(CSRC)/mcx_fastb.hpp

mcpar_amd/libmcx.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f mcpar_amd/libmcx.so $(CSRC)/*.o $(CSRC)/mcx_rtc_headers.inc
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean

drivers: lib
	$(MAKE) -C mcpar_amd/drivers

test-cpu: lib oracle
	python -m pytest tests -x -q -m "not gpu"

.PHONY: drivers test-cpu
