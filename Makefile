# Top-level conveniences.  The product is built by video-stab_amd/csrc/Makefile (hipcc, gfx950), the CPU oracle by
# oracle/Makefile; `python -c "import __graft_entry__ as g; g.build()"` builds both.
#
#   make            product library + oracle + test programs
#   make asan       CPU-only sanitizer run (no GPU needed, none used): the oracle and the product's HOST code under
#                   AddressSanitizer + UndefinedBehaviorSanitizer -
#                     * the oracle library rebuilt with -fsanitize=address,undefined and driven by its known-answer tests,
#                       the roll / zoom-crop / enhancer oracle tests and the libm restatement check;
#                     * the product's host-only translation units (config reader, AutoZoomCrop contour logic) in their
#                       fuzz harnesses (scratch/fuzz): 20 000 mutated config documents, 5 000 random masks;
#                     * vs_libm.h (host build) over 2^24 arguments per function against the host libm.
#                   GPU sanitizers are not available on this pool; device code is covered by the parity tests instead.
SAN := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
ASAN_LIB := $(shell gcc -print-file-name=libasan.so)

all:
	$(MAKE) -C video-stab_amd/csrc
	$(MAKE) -C oracle
	$(MAKE) -C tests/cpp

asan: asan-oracle asan-host
	@echo "make asan: no sanitizer report"

asan-oracle:
	$(MAKE) -C oracle OUT=_asan/libvso_oracle.so CXXFLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread $(SAN)"
	LD_PRELOAD=$(ASAN_LIB) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
	  VSO_ORACLE_LIB=$(CURDIR)/oracle/_asan/libvso_oracle.so \
	  python3 -m pytest tests/test_oracle_kat.py tests/test_roll.py tests/test_azc.py tests/test_enhance.py -q -x -m "not gpu" -p no:cacheprovider

asan-host:
	@mkdir -p scratch/fuzz/_asan
	g++ -O1 -g -std=c++17 $(SAN) -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -I video-stab_amd/csrc \
	    scratch/fuzz/config_fuzz.cpp scratch/fuzz/stubs.cpp video-stab_amd/csrc/config.cpp -o scratch/fuzz/_asan/config_fuzz
	g++ -O1 -g -std=c++17 $(SAN) -I include -I video-stab_amd/csrc \
	    scratch/fuzz/azc_fuzz.cpp scratch/fuzz/azc_stubs.cpp video-stab_amd/csrc/azc_contour.cpp -o scratch/fuzz/_asan/azc_fuzz
	g++ -O1 -g -std=c++17 -ffp-contract=off -pthread $(SAN) -DLIBM_CHECK_QUICK tests/cpp/libm_check.cpp -o scratch/fuzz/_asan/libm_check
	ASAN_OPTIONS=detect_leaks=1 ./scratch/fuzz/_asan/config_fuzz 20000
	ASAN_OPTIONS=detect_leaks=1 ./scratch/fuzz/_asan/azc_fuzz 5000
	ASAN_OPTIONS=detect_leaks=1 ./scratch/fuzz/_asan/libm_check quick

.PHONY: all asan asan-oracle asan-host
