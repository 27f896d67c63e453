# Builds clip_cpp_amd/libclip.so (the drop-in C-ABI library: include/clip.h + include/clip_amd.h) and the stub libggml.so
# for C / C++ consumers without Python (reference: CMakeLists.txt:255-272 builds `clip` + `ggml`).
#   make            -> clip_cpp_amd/libclip.so clip_cpp_amd/libggml.so
#   make hooks=0    -> without the kernel test / micro-benchmark hooks (clip_amd_test_*, clip_amd_bench_gemm)
#   make example    -> build/simple: a minimal C caller linked against the library
# gfx950 only.  `python -m clip_cpp_amd.build` does the same thing incrementally (and is what __graft_entry__.build() runs).
HIPCC   ?= $(shell command -v hipcc || echo /opt/rocm/bin/hipcc)
ARCH    ?= gfx950
hooks   ?= 1
SRC     := clip_cpp_amd/csrc
OUT     := clip_cpp_amd/build
CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result -Wno-inline-asm -Wno-bitwise-instead-of-logical -Iinclude --offload-arch=$(ARCH) -DCLIPAMD_TEST_HOOKS=$(hooks)
HOST    := gguf quant load forward tokenizer preprocess image_io image_formats jpeg_decode host_pipeline api
KERNELS := k_attn k_attn_f32 k_misc k_preproc k_gemm k_gemm8 k_gemm4 k_gemm32 k_gemm_f32 k_skinny k_gemm_ring k_fold
WTS     := 0 1 2 3 4 5
OBJS    := $(HOST:%=$(OUT)/%.cpp.o) $(KERNELS:%=$(OUT)/%.hip.o) $(WTS:%=$(OUT)/k_gemm_wt%.o) $(WTS:%=$(OUT)/k_skinny_wt%.o) $(WTS:%=$(OUT)/k_gemm_ring_wt%.o)

all: clip_cpp_amd/libclip.so clip_cpp_amd/libggml.so

$(OUT):
	mkdir -p $(OUT)
# (-fwrapv: the file decoders run integer transforms over untrusted coefficients; same flag as clip_cpp_amd/build.py)
$(OUT)/%.cpp.o: $(SRC)/%.cpp $(wildcard $(SRC)/*.h) include/clip.h include/clip_amd.h | $(OUT)
	$(HIPCC) -x hip -fwrapv $(CXXFLAGS) -c $< -o $@
# (the f32-file kernels without the SLP vectoriser: same per-file flags as clip_cpp_amd/build.py EXTRA_FLAGS, which says why)
$(OUT)/k_attn_f32.hip.o $(OUT)/k_gemm_f32.hip.o: CXXFLAGS += -fno-slp-vectorize
$(OUT)/%.hip.o: $(SRC)/%.hip $(wildcard $(SRC)/*.h) | $(OUT)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@
$(OUT)/k_gemm_wt%.o: $(SRC)/k_gemm.hip $(wildcard $(SRC)/*.h) | $(OUT)
	$(HIPCC) $(CXXFLAGS) -DCLIPAMD_GEMM_WT=$* -c $< -o $@
$(OUT)/k_skinny_wt%.o: $(SRC)/k_skinny.hip $(wildcard $(SRC)/*.h) | $(OUT)
	$(HIPCC) $(CXXFLAGS) -DCLIPAMD_SKINNY_WT=$* -c $< -o $@
$(OUT)/k_gemm_ring_wt%.o: $(SRC)/k_gemm_ring.hip $(wildcard $(SRC)/*.h) | $(OUT)
	$(HIPCC) $(CXXFLAGS) -DCLIPAMD_RING_WT=$* -c $< -o $@
clip_cpp_amd/libclip.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lz -lpthread -ldl
clip_cpp_amd/libggml.so: $(SRC)/ggml_stub.c
	gcc -O2 -fPIC -shared -o $@ $<
example: all
	mkdir -p build && gcc -O2 -Iinclude -o build/simple /root/reference/examples/simple.c -Lclip_cpp_amd -lclip -Wl,-rpath,'$$ORIGIN/../clip_cpp_amd' 2>/dev/null || \
	  echo "reference examples/simple.c not available here: see INTEGRATION.md for the caller"
clean:
	rm -rf $(OUT) clip_cpp_amd/libclip.so clip_cpp_amd/libggml.so
.PHONY: all example clean
