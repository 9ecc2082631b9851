# Builds libcffm_hip.so (gfx950 only), the host-side libfm reader and the pybind11 layer over the C ABI.  hipcc cross-compiles
# without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := cffm_amd/csrc
OUT   := cffm_amd/lib
OBJS  := $(patsubst $(SRC)/%.hip,build/%.o,$(wildcard $(SRC)/*.hip))
CXXFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -Wall -Wno-unused-function -Wno-unused-variable
ifdef PHASE_TIMERS
CXXFLAGS += -DCFFM_PHASE_TIMERS
endif
# debug build only: CFFM_DBG=<bits> skips phases of the tiled layer-0 kernels (tools/dbg_tile.py)
ifdef TILE_DBG
CXXFLAGS += -DCFFM_TILE_DBG
endif

PYEXT := $(OUT)/_cffm_pybind$(shell python3-config --extension-suffix)

all: $(OUT)/libcffm_hip.so $(OUT)/libcffm_libfm.so $(PYEXT)

# thin pybind11 layer over the C ABI (north_star: "through a thin pybind11 C-ABI layer"); links the library next to it
$(PYEXT): cffm_amd/csrc_host/pybind_module.cpp include/cffm_hip.h $(OUT)/libcffm_hip.so
	g++ -O2 -std=c++17 -fPIC -shared -fvisibility=hidden $(shell python3 -m pybind11 --includes) $< -o $@ -L$(OUT) -lcffm_hip -Wl,-rpath,'$$ORIGIN' \
	  || echo 'WARNING: the pybind11 layer did not build (pybind11 / Python headers missing?); cffm_amd.hip.fast() falls back to ctypes'

# host-only fast libfm reader
$(OUT)/libcffm_libfm.so: cffm_amd/csrc_host/libfm_reader.cpp
	@mkdir -p $(OUT)
	g++ -O2 -std=c++17 -fPIC -shared -Wall $< -o $@

build/%.o: $(SRC)/%.hip $(wildcard $(SRC)/*.hpp) include/cffm_hip.h
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(OUT)/libcffm_hip.so: $(OBJS)
	@mkdir -p $(OUT)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

# CPU sanitizer build of the two host shims (libfm reader, pybind11 layer) and the loader / ABI tests run against it.
# CPU only: GPU AddressSanitizer is not available on the MI355X pool.
ASAN_DIR := build/asan
ASAN_FLAGS := -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined
asan: $(OUT)/libcffm_hip.so
	@mkdir -p $(ASAN_DIR)
	g++ $(ASAN_FLAGS) -Wall cffm_amd/csrc_host/libfm_reader.cpp -o $(ASAN_DIR)/libcffm_libfm.so
	g++ $(ASAN_FLAGS) -fvisibility=hidden $(shell python3 -m pybind11 --includes) cffm_amd/csrc_host/pybind_module.cpp \
	  -o $(ASAN_DIR)/_cffm_pybind$(shell python3-config --extension-suffix) -L$(OUT) -lcffm_hip -Wl,-rpath,'$(abspath $(OUT))'
	LD_PRELOAD="$(shell gcc -print-file-name=libasan.so) $(shell gcc -print-file-name=libubsan.so)" \
	  ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
	  CFFM_HOST_LIB_DIR=$(abspath $(ASAN_DIR)) python3 -m pytest tests/test_loader.py tests/test_abi.py -x -q -m "not gpu" -p no:cacheprovider

clean:
	rm -rf build $(OUT)/libcffm_hip.so $(OUT)/libcffm_libfm.so $(PYEXT)
.PHONY: all clean asan
