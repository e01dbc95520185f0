# Builds the MI355X SSIM engine (gfx950 only) and the checker libraries.
#
#   make            -> ssim_amd/lib/librmgr-ssim-hip.so  + oracle libs
#   make lib        -> the product library only
#   make oracle     -> oracle/libssim_oracle.so (+ oracle/_ref when /root/reference exists)
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
SRC     := ssim_amd/csrc
OUT     := ssim_amd/lib
OBJ     := build/obj
BIN     := ssim_amd/bin
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(SRC) -Wall -Wno-unused-function

all: lib oracle

lib: $(OUT)/librmgr-ssim-hip.so $(BIN)/rmgr-ssim

$(OBJ)/ssim_kernels.o: $(SRC)/ssim_kernels.hip $(SRC)/ssim_kernels.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/ssim_hip_abi.o: $(SRC)/ssim_hip_abi.cpp $(SRC)/ssim_kernels.h include/rmgr/ssim-hip.h include/rmgr/ssim.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

# The drop-in layer is plain C++98 compiled by the host compiler: no HIP on this side.
$(OBJ)/ssim_dropin.o: $(SRC)/ssim_dropin.cpp include/rmgr/ssim.h include/rmgr/ssim-openmp.h include/rmgr/ssim-hip.h include/rmgr/ssim-version.h
	@mkdir -p $(OBJ)
	$(CXX) -std=c++98 -pedantic -O2 -fPIC -Wall -Wextra -Iinclude -c $< -o $@

$(OUT)/librmgr-ssim-hip.so: $(OBJ)/ssim_kernels.o $(OBJ)/ssim_hip_abi.o $(OBJ)/ssim_dropin.o
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

# The command-line tool: plain host C++ on top of the C ABI (reference: src/ssim-cli.cpp).
$(BIN)/rmgr-ssim: $(SRC)/ssim_cli.cpp $(OUT)/librmgr-ssim-hip.so include/rmgr/ssim.h include/rmgr/ssim-hip.h
	@mkdir -p $(BIN)
	$(CXX) -std=c++98 -O2 -Wall -Wextra -Iinclude $< -o $@ -L$(OUT) -lrmgr-ssim-hip -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf build $(OUT) $(BIN)
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean
