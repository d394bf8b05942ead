# Builds liblstc_hip.so (gfx950 only) and the oracle-side helpers.  `python -c "import __graft_entry__ as g; g.build()"`
# drives the same commands.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := lstc_vad_amd/csrc
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(SRCS:.hip=.o)
LIB   := lstc_vad_amd/liblstc_hip.so
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Werror=return-type -Wno-unused-function -ffp-contract=off $(EXTRA_HIPFLAGS)

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/lstc_common.h $(CSRC)/attention_common.h include/lstc_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS)

tools/gemm_check: tools/gemm_check.cpp $(LIB)
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -Iinclude $< -o $@ -Llstc_vad_amd -llstc_hip -Wl,-rpath,'$$ORIGIN/../lstc_vad_amd'

# tuning build (timing ablations compiled in; NEVER the product): tools/tuning/liblstc_hip.so + tools/tuning/gemm_check
TOBJS := $(patsubst $(CSRC)/%.hip,tools/tuning/%.o,$(SRCS))
tools/tuning/%.o: $(CSRC)/%.hip $(CSRC)/lstc_common.h $(CSRC)/attention_common.h include/lstc_hip.h
	@mkdir -p tools/tuning
	$(HIPCC) $(HIPFLAGS) -DLSTC_TUNING -c $< -o $@
tools/tuning/liblstc_hip.so: $(TOBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(TOBJS)
tools/tuning/gemm_check: tools/gemm_check.cpp tools/tuning/liblstc_hip.so
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -DLSTC_TUNING -Iinclude $< -o $@ -Ltools/tuning -llstc_hip -Wl,-rpath,'$$ORIGIN'
tuning: tools/tuning/gemm_check

clean:
	rm -f $(OBJS) $(LIB) tools/gemm_check

.PHONY: all clean tuning
