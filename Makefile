# Builds libsac_hip.so (the product) for gfx950 and the CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
HIPFLAGS ?= -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fPIC -Wall -Wno-unused-value
SRC = sac_amd/csrc/sac_hip.hip sac_amd/csrc/sac_edge.hip
HDR = $(sort $(wildcard sac_amd/csrc/*.h)) include/sac_hip.h
# fingerprint of the sources a library was built from (sac_build_id()): ties rocprofv3 counter files (profiles/traffic.json)
# to the binary they were measured on
BUILD_ID := $(shell cat $(SRC) $(HDR) | sha256sum | cut -c1-16)
HIPFLAGS += -DSAC_BUILD_ID=\"$(BUILD_ID)\"

all: sac_amd/libsac_hip.so oracle

sac_amd/libsac_hip.so: $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRC) -ldl

asm: $(SRC) $(HDR)
	mkdir -p build && $(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/sac_hip.s sac_amd/csrc/sac_hip.hip -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -f sac_amd/libsac_hip.so; rm -rf build; $(MAKE) -C oracle clean

.PHONY: all oracle clean asm

probe: tools/gather_probe tools/lone_wave_probe
tools/lone_wave_probe: tools/lone_wave_probe.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -Wno-unused-result -o $@ $<
tools/gather_probe: tools/gather_probe.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -o $@ $<
