# Builds the MI355X SSIM engine (gfx950 only) and the checker libraries.
#
#   make            -> ssim_amd/lib/librmgr-ssim-hip.so  + oracle libs
#   make lib        -> the product library only
#   make oracle     -> oracle/libssim_oracle.so (+ oracle/_ref when /root/reference exists)
#   make sanitize   -> CPU-only ASan/UBSan pass over the tool's codecs and the oracle
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
SRC     := ssim_amd/csrc
OUT     := ssim_amd/lib
OBJ     := build/obj
BIN     := ssim_amd/bin
# DOUBLE=1 builds the library with the reference's RMGR_SSIM_USE_DOUBLE semantics as the default mode
# (CMakeLists.txt:53 of the reference); at run time RMGR_SSIM_HIP_MODE or rmgr_ssim_hip_set_mode override it.
HIPFLAGS := $(if $(DOUBLE),-DRMGR_SSIM_USE_DOUBLE=1) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(SRC) -Wall -Wno-unused-function

all: lib oracle

lib: $(OUT)/librmgr-ssim-hip.so $(OUT)/librmgr-ssim-hip-double.so $(OUT)/librmgr-ssim.a $(OUT)/librmgr-ssim-openmp.a $(BIN)/rmgr-ssim

# Static flavour under the reference's archive name (CMakeLists.txt:205): the same three objects, linked into ONE relocatable
# object whose only global symbols are the API (the shared libraries' export list has no counterpart for archives: the
# internal ssim_hip:: interface between the ABI layer and the kernels would otherwise be visible to -- and collide with --
# whatever else the program links; the reference's archive exposes its API only).  A program that links it also needs the
# HIP runtime: g++ app.o -lrmgr-ssim -L/opt/rocm/lib -lamdhip64 -ldl -lpthread
# --force-group-allocation: the objects carry COMDAT groups (inline functions, typeinfo, guard variables of the C++ runtime
# headers).  A plain `ld -r` keeps the groups and objcopy below then makes their symbols LOCAL; a client object that carries
# a group of the same signature (anything using std::make_shared, std::thread, ...) makes the final link discard the
# archive's copy and leaves its now-local references dangling ("defined in discarded section").  Allocated into ordinary
# sections here, the archive's copies are private to it and always kept.
$(OUT)/librmgr-ssim.a: $(OBJ)/ssim_kernels.o $(OBJ)/ssim_probe.o $(OBJ)/ssim_hip_abi.o $(OBJ)/ssim_dropin.o
	@mkdir -p $(OUT)
	ld -r --force-group-allocation -o $(OBJ)/rmgr_ssim_api.o $^
	objcopy --wildcard --keep-global-symbol='rmgr_ssim_*' --keep-global-symbol='_ZN4rmgr4ssim12compute_ssimE*' --keep-global-symbol='_ZN4rmgr4ssim11select_implE*' $(OBJ)/rmgr_ssim_api.o
	rm -f $@ && ar rcs $@ $(OBJ)/rmgr_ssim_api.o

# The reference's second archive (CMakeLists.txt:229: rmgr-ssim-openmp = src/ssim-openmp.c): rmgr_ssim_compute_ssim_openmp() and
# nothing else, resolved against librmgr-ssim like the reference's (-lrmgr-ssim-openmp -lrmgr-ssim); the shared libraries carry
# the same object.
$(OUT)/librmgr-ssim-openmp.a: $(OBJ)/ssim_openmp.o
	@mkdir -p $(OUT)
	rm -f $@ && ar rcs $@ $^

$(OBJ)/ssim_openmp.o: $(SRC)/ssim_openmp.c include/rmgr/ssim-openmp.h include/rmgr/ssim.h
	@mkdir -p $(OBJ)
	$(CC) -std=c89 -pedantic -O2 -fPIC -Wall -Wextra -Iinclude -c $< -o $@

# The kernels carry the sha256 of their own source (rmgr_ssim_hip_get_kernel_source_id): measurements that belong to one version of
# the kernels -- profiles/traffic.json -- name it, and bench.py refuses to quote them for any other.
$(OBJ)/ssim_kernels.o: $(SRC)/ssim_kernels.hip $(SRC)/ssim_kernels.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DSSIM_KERNELS_SOURCE_ID=\"$$(sha256sum $< | cut -c1-64)\" -c $< -o $@

# The forced-occupancy VALU probe behind rmgr_ssim_hip_probe_valu (profiling aid): its own file, so that the kernel source id above names the SSIM kernels only.
$(OBJ)/ssim_probe.o: $(SRC)/ssim_probe.hip $(SRC)/ssim_kernels.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/ssim_hip_abi.o: $(SRC)/ssim_hip_abi.cpp $(SRC)/ssim_kernels.h include/rmgr/ssim-hip.h include/rmgr/ssim.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

# The drop-in layer is plain C++98 compiled by the host compiler: no HIP on this side.
$(OBJ)/ssim_dropin.o: $(SRC)/ssim_dropin.cpp $(SRC)/ssim_internal.h include/rmgr/ssim.h include/rmgr/ssim-openmp.h include/rmgr/ssim-hip.h include/rmgr/ssim-version.h
	@mkdir -p $(OBJ)
	$(CXX) -std=c++98 -pedantic -O2 -fPIC -Wall -Wextra -Iinclude -c $< -o $@

# Only the API leaves the shared libraries: $(SRC)/exports.map (the reference's archive exposes only its API as well).
EXPORTS := -Wl,--version-script=$(SRC)/exports.map

$(OUT)/librmgr-ssim-hip.so: $(OBJ)/ssim_kernels.o $(OBJ)/ssim_probe.o $(OBJ)/ssim_hip_abi.o $(OBJ)/ssim_dropin.o $(OBJ)/ssim_openmp.o $(SRC)/exports.map
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(EXPORTS) -o $@ $(filter %.o,$^)

# The reference's RMGR_SSIM_USE_DOUBLE build configuration (CMakeLists.txt:53, src/ssim_internal.h:26-37) as a second
# flavour of the same library: identical kernels and drop-in layer, only the C ABI's default arithmetic differs (fp64
# internals for every unchanged rmgr_ssim_compute_ssim call; BASELINE.json configs[4]).  `make DOUBLE=1` gives the same
# thing under the main name.
$(OBJ)/ssim_hip_abi_double.o: $(SRC)/ssim_hip_abi.cpp $(SRC)/ssim_kernels.h include/rmgr/ssim-hip.h include/rmgr/ssim.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DRMGR_SSIM_USE_DOUBLE=1 -x hip -c $< -o $@

$(OUT)/librmgr-ssim-hip-double.so: $(OBJ)/ssim_kernels.o $(OBJ)/ssim_probe.o $(OBJ)/ssim_hip_abi_double.o $(OBJ)/ssim_dropin.o $(OBJ)/ssim_openmp.o $(SRC)/exports.map
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(EXPORTS) -o $@ $(filter %.o,$^)

# The command-line tool: plain host C++ on top of the C ABI (reference: src/ssim-cli.cpp).
$(BIN)/rmgr-ssim: $(SRC)/ssim_cli.cpp $(OUT)/librmgr-ssim-hip.so include/rmgr/ssim.h include/rmgr/ssim-hip.h
	@mkdir -p $(BIN)
	$(CXX) -std=c++98 -O2 -Wall -Wextra -Iinclude $< -o $@ -L$(OUT) -lrmgr-ssim-hip -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle all

# Install layout of the reference (CMakeLists.txt:321-338): headers under include/rmgr, libraries under lib.
# The reference's two link names (-lrmgr-ssim, -lrmgr-ssim-openmp) resolve to the one HIP library.
PREFIX ?= /usr/local
install: lib
	install -d $(DESTDIR)$(PREFIX)/include/rmgr $(DESTDIR)$(PREFIX)/lib $(DESTDIR)$(PREFIX)/bin $(DESTDIR)$(PREFIX)/lib/pkgconfig
	install -m 644 include/rmgr/ssim.h include/rmgr/ssim-openmp.h include/rmgr/ssim-version.h include/rmgr/ssim-hip.h $(DESTDIR)$(PREFIX)/include/rmgr/
	install -m 755 $(OUT)/librmgr-ssim-hip.so $(OUT)/librmgr-ssim-hip-double.so $(DESTDIR)$(PREFIX)/lib/
	install -m 644 $(OUT)/librmgr-ssim.a $(OUT)/librmgr-ssim-openmp.a $(DESTDIR)$(PREFIX)/lib/
	ln -sf librmgr-ssim-hip.so $(DESTDIR)$(PREFIX)/lib/librmgr-ssim.so
	ln -sf librmgr-ssim-hip.so $(DESTDIR)$(PREFIX)/lib/librmgr-ssim-openmp.so
	install -m 755 $(BIN)/rmgr-ssim $(DESTDIR)$(PREFIX)/bin/
	printf 'prefix=%s\nlibdir=$${prefix}/lib\nincludedir=$${prefix}/include\n\nName: rmgr-ssim\nDescription: SSIM (rmgr::ssim API) on AMD MI355X / gfx950\nVersion: 2.1.0\nLibs: -L$${libdir} -lrmgr-ssim-hip\nCflags: -I$${includedir}\n' '$(PREFIX)' > $(DESTDIR)$(PREFIX)/lib/pkgconfig/rmgr-ssim.pc

# CPU-only sanitizer pass (GPU ASan is not available on this pool): the command-line tool's own PNG / JPEG / PNM / BMP / TGA
# readers and writers against valid, odd and mutated files (tests/test_cli.py, no GPU needed for --decode), and the oracle's
# C restatement, both built with -fsanitize=address,undefined.
sanitize: lib oracle
	@mkdir -p build/sanitize
	$(CXX) -std=c++98 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -Wextra -Iinclude $(SRC)/ssim_cli.cpp -o build/sanitize/rmgr-ssim-asan -L$(OUT) -lrmgr-ssim-hip -Wl,-rpath,$(abspath $(OUT))
	ASAN_OPTIONS=detect_leaks=0 RMGR_SSIM_CLI=$(abspath build/sanitize/rmgr-ssim-asan) python3 -m pytest tests/test_cli.py -x -q -m "not gpu"
	$(CC) -O1 -g -std=c99 -D_GNU_SOURCE -ffp-contract=off -fno-math-errno -fopenmp -mavx2 -mfma -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -shared -fPIC -o build/sanitize/libssim_oracle_asan.so oracle/ssim_oracle.c -lm
	LD_PRELOAD=$$($(CC) -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 tests/tools/sanitize_oracle.py $(abspath build/sanitize/libssim_oracle_asan.so) $(abspath .)

clean:
	rm -rf build $(OUT) $(BIN)
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean install sanitize
