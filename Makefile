# Builds libcffm_hip.so (gfx950 only) and the oracle's compiled helpers.  hipcc cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := cffm_amd/csrc
OUT   := cffm_amd/lib
OBJS  := $(patsubst $(SRC)/%.hip,build/%.o,$(wildcard $(SRC)/*.hip))
CXXFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -Wall -Wno-unused-function -Wno-unused-variable

all: $(OUT)/libcffm_hip.so

build/%.o: $(SRC)/%.hip $(SRC)/common.hpp include/cffm_hip.h
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(OUT)/libcffm_hip.so: $(OBJS)
	@mkdir -p $(OUT)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

clean:
	rm -rf build $(OUT)/libcffm_hip.so
.PHONY: all clean
