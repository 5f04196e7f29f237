# Top-level build: the HIP library (the product), the C host program, the oracle (test infrastructure)
# and the microbenchmark.  gfx950 only.  `python -c "import __graft_entry__ as g; g.build()"` runs this.
HIPCC   ?= hipcc
CC      ?= gcc
ARCH    ?= gfx950
PKG     := mini_nbody_amd
CSRC    := $(PKG)/csrc
# -ffp-contract=off: products are fused only where the source says fma (rounding points are part of the contract)
# -fno-slp-vectorize: v_pk_*_f32 costs 4 cycles on gfx950 (profiles/r01_microbench_valu_issue.txt): no gain, more registers
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function

all: lib host oracle microbench

# The library = ONE device translation unit (kernels.hip: the nbk kernels + the launch functions that pick an instantiation; ~45 s of hipcc)
# and three host-only C++ files (context, comm, mailbox: seconds each) behind csrc/nbody_internal.hpp.  A host-side edit relinks in seconds.
HOSTFLAGS := -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function
OBJ      := build/obj
KERNEL_SRC := $(CSRC)/kernels.hip $(CSRC)/nbody_kernels.hpp $(CSRC)/nbody_args.hpp $(CSRC)/force_loop_gfx950.inc
HOST_HDR := $(CSRC)/nbody_internal.hpp $(CSRC)/nbody_args.hpp include/nbody.h
HOST_OBJ := $(OBJ)/context.o $(OBJ)/comm.o $(OBJ)/mailbox.o

lib: $(PKG)/libnbody_hip.so
$(OBJ)/kernels.o: $(KERNEL_SRC) $(HOST_HDR)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $(CSRC)/kernels.hip -o $@
$(OBJ)/%.o: $(CSRC)/%.cpp $(HOST_HDR)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HOSTFLAGS) -c $< -o $@
$(PKG)/libnbody_hip.so: $(OBJ)/kernels.o $(HOST_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -pthread -o $@ $^ -ldl

# The diagnostic library: the same sources with kernels.hip built -DNBODY_DIAG_LOOPS — the experiment encodings of the hand-scheduled loop
# and its TIMING-ONLY forms (wrong results) that profiles/r02_loop_diagnostics.md was measured with.  Not part of `all`, never loaded by
# the package unless NBODY_LIB points at it (tools/profile_diag.sh does).  The host objects are the product's own.
diag: $(PKG)/libnbody_hip_diag.so
$(OBJ)/kernels_diag.o: $(KERNEL_SRC) $(HOST_HDR)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DNBODY_DIAG_LOOPS -c $(CSRC)/kernels.hip -o $@
$(PKG)/libnbody_hip_diag.so: $(OBJ)/kernels_diag.o $(HOST_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -pthread -o $@ $^ -ldl

# C host program (north_star: "host code stays in C"): links only the C-ABI
host: build/nbody build/mailbox_driver
build/nbody: $(PKG)/host/nbody.c include/nbody.h include/nbody_ic.h $(PKG)/libnbody_hip.so
	@mkdir -p build
	$(CC) -std=c11 -O2 -Wall -Iinclude -o $@ $(PKG)/host/nbody.c -L$(PKG) -lnbody_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)' -Wl,-rpath,/opt/rocm/lib -lm

# the PS-side driver of the reference's mailbox, plain C over the same C-ABI
build/mailbox_driver: $(PKG)/host/mailbox_driver.c include/nbody.h include/nbody_ic.h $(PKG)/libnbody_hip.so
	@mkdir -p build
	$(CC) -std=c11 -O2 -Wall -Iinclude -o $@ $(PKG)/host/mailbox_driver.c -L$(PKG) -lnbody_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)' -Wl,-rpath,/opt/rocm/lib -lm

oracle:
	$(MAKE) -C oracle

microbench: build/microbench build/microbench_streams build/microbench_roles build/microbench_mfma build/microbench_gridsync
build/microbench: $(CSRC)/microbench.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
build/microbench_roles: $(CSRC)/microbench_roles.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
# 0.5 MB of generated instruction streams: produced on demand, not tracked
$(CSRC)/microbench_streams.inc: tools/gen_streams.py tools/gen_force_loop.py
	python3 tools/gen_streams.py
build/microbench_streams: $(CSRC)/microbench_streams.hip $(CSRC)/microbench_streams.inc
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<

# round 4's probe: can the matrix pipe carry the coordinate differences? (no: profiles/r04_mfma_differences.md); its loop is generated
$(CSRC)/force_loop_mfma_gfx950.inc: tools/gen_mfma_loop.py
	python3 tools/gen_mfma_loop.py
build/microbench_mfma: $(CSRC)/microbench_mfma.hip $(CSRC)/force_loop_mfma_gfx950.inc
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -ffp-contract=off -fno-slp-vectorize -o $@ $<

# round 4: what a step boundary costs inside a launch against the kernel boundary (profiles/r04_step_boundary.md)
build/microbench_gridsync: $(CSRC)/microbench_gridsync.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<

# generated sources: the hand-scheduled loop (committed) and the microbenchmark streams (not tracked)
gen:
	python3 tools/gen_force_loop.py
	python3 tools/gen_streams.py

isa: $(KERNEL_SRC)
	@mkdir -p build/isa
	cd build/isa && $(HIPCC) $(HIPFLAGS) -c ../../$(CSRC)/kernels.hip -save-temps -Rpass-analysis=kernel-resource-usage -o kernels.o 2> resource_usage.txt

clean:
	rm -rf build $(PKG)/libnbody_hip.so $(PKG)/libnbody_hip_diag.so
	$(MAKE) -C oracle clean

.PHONY: all lib diag host oracle microbench isa gen clean
