# Top-level build of the C-ABI engine and the drop-in CLI drivers (gfx950 only).
#   make            -> gunrock_amd/libgrx.so + bin/{bfs,sssp,pr}
#   make header_only-> bin/*_generic: same drivers on the generic operators only (no libgrx)
HIPCC ?= /opt/rocm/bin/hipcc
FLAGS = -std=c++17 -O3 --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Iinclude -x hip
LINK = -Lgunrock_amd -lgrx -Wl,-rpath,'$$ORIGIN/../gunrock_amd'
HDRS = $(shell find include -name '*.hxx' -o -name '*.h') $(wildcard examples/algorithms/*.hxx examples/algorithms/*/*.hxx)

all: lib bin/bfs bin/sssp bin/pr bin/test_engine_cache
lib:
	python -m gunrock_amd.build
bin/%: examples/algorithms/%/*.cu $(HDRS) | lib
	@mkdir -p bin
	$(HIPCC) $(FLAGS) $< -o $@ $(LINK)
bin/test_engine_cache: tests/cpp/test_engine_cache.cu $(HDRS) | lib
	@mkdir -p bin
	$(HIPCC) $(FLAGS) $< -o $@ $(LINK)
bin/test_operators: tests/cpp/test_operators.cu $(HDRS)
	@mkdir -p bin
	$(HIPCC) $(FLAGS) -DGUNROCK_HEADER_ONLY $< -o $@
bin/test_host_utils: tests/cpp/test_host_utils.cu $(HDRS)
	@mkdir -p bin
	$(HIPCC) $(FLAGS) $< -o $@ -lpthread
header_only: bin/bfs_generic bin/sssp_generic bin/pr_generic bin/test_operators bin/test_host_utils
bin/%_generic: examples/algorithms/%/*.cu $(HDRS)
	@mkdir -p bin
	$(HIPCC) $(FLAGS) -DGUNROCK_HEADER_ONLY $< -o $@
clean:
	rm -rf bin gunrock_amd/_obj gunrock_amd/libgrx.so
.PHONY: all lib header_only clean
