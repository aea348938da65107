# Builds the product library (HIP kernels + C ABI) for gfx950, the headless host
# tools, and (test infrastructure) the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := path-tracer_amd
CSRC := $(PKG)/csrc
LIB := $(PKG)/lib/libklrender.so
# -ffp-contract=off: the kernels restate reference arithmetic operation by operation and spell
# out every fused multiply-add they want; the compiler must not add or remove any.
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math \
            -fvisibility=hidden -Wall -Wno-unused-function -Iinclude -I$(CSRC)
SRCS := $(CSRC)/kl_kernels.hip $(CSRC)/kl_lbvh.hip $(CSRC)/kl_group.hip $(CSRC)/kl_api.cpp $(CSRC)/kl_bvh.cpp $(CSRC)/kl_scene.cpp $(CSRC)/kl_mesh_io.cpp
OBJS := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.hpp) include/Math/kl_render.h

all: $(LIB) tools oracle

$(LIB): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -lpthread -ldl

build/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
build/%.cpp.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

tools: $(LIB)
	@if [ -f $(PKG)/host/kl_bench.cpp ]; then mkdir -p $(PKG)/bin && \
	  g++ -O2 -std=c++17 -ffp-contract=off -Iinclude -I$(PKG)/host -o $(PKG)/bin/kl_bench $(PKG)/host/kl_bench.cpp $(PKG)/host/*_impl.cpp \
	      -L$(PKG)/lib -lklrender -Wl,-rpath,'$$ORIGIN/../lib' -lz ; fi

oracle:
	$(MAKE) -C oracle all

# Host-side code that handles caller data (mesh and image loaders, builders), and the oracle, under AddressSanitizer + UBSan (CPU only: g++/gcc, no device code),
# run over well-formed, truncated and garbage mesh and image files (tools/asan/run.py).
ASAN_FLAGS := -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined
asan:
	@mkdir -p build/asan
	g++ $(ASAN_FLAGS) -std=c++17 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -I$(CSRC) -o build/asan/host_check \
	    tools/asan/host_check.cpp $(CSRC)/kl_mesh_io.cpp $(CSRC)/kl_bvh.cpp $(CSRC)/kl_scene.cpp -lpthread
	g++ $(ASAN_FLAGS) -std=c++17 -I$(PKG)/host -o build/asan/image_check tools/asan/image_check.cpp $(PKG)/host/kl_image_impl.cpp $(PKG)/host/kl_jpeg_impl.cpp -lz
	gcc $(ASAN_FLAGS) -std=c11 -Wno-stringop-overread -fopenmp -mfma -ffp-contract=off -Ioracle -o build/asan/oracle_check tools/asan/oracle_check.c oracle/kl_oracle.c -lm
	python3 tools/asan/run.py

clean:
	rm -rf build $(PKG)/lib $(PKG)/bin; $(MAKE) -C oracle clean

.PHONY: all tools oracle clean asan
