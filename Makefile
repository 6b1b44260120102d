# Convenience targets; the python entry points do the same (helib_amd/build.py, __graft_entry__.build()).
#   make lib        hipcc --offload-arch=gfx950 -> helib_amd/lib/libhelib_amd.so (cross-compiles without a GPU)
#   make oracle     the CPU restatement used as the checker (test infrastructure)
#   make test-cpu   pytest -m "not gpu"      make test-gpu   pytest -m gpu (needs an MI355X)
#   make bench      python bench.py          (one JSON line; --gpus N spawns the ranks)
HIPCC ?= $(shell command -v hipcc 2>/dev/null || echo /opt/rocm/bin/hipcc)
HIPFLAGS = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed
CSRC = helib_amd/csrc
LIB = helib_amd/lib
HDRS = $(wildcard $(CSRC)/*.h) include/helib_amd.h

OBJS = $(LIB)/ntt_kernels_13.o $(LIB)/ntt_kernels_14.o $(LIB)/ntt_kernels_15.o $(LIB)/ntt_dispatch.o $(LIB)/conv_kernels.o $(LIB)/pfa_kernels.o $(LIB)/rns_mfma_kernels.o $(LIB)/engine.o
lib: $(LIB)/libhelib_amd.so
$(LIB)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
# the row kernels: one translation unit per ring size (make -j builds them in parallel)
$(LIB)/ntt_kernels_%.o: $(CSRC)/ntt_kernels.hip $(HDRS)
	@mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) -DHX_NTT_ONLY=$* -c $< -o $@
$(LIB)/libhelib_amd.so: $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^
oracle:
	$(MAKE) -s -C oracle
test-cpu: lib oracle
	python -m pytest tests -q -m "not gpu"
test-gpu: lib oracle
	python -m pytest tests -q -m gpu
bench: lib
	python bench.py
clean:
	rm -f $(LIB)/*.o $(LIB)/libhelib_amd.so
.PHONY: lib oracle test-cpu test-gpu bench clean
