# Builds the HIP libraries (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
#   cmr_agent_amd/lib/libcmr_hip.so      the product: include/cmr_hip.h, no mutable state
#   cmr_agent_amd/lib/libcmr_hip_ab.so   the same sources with -DCMR_AB_SWITCHES: + the kernel-variant switches of include/cmr_hip_ab.h
#                                        (tests that compare two kernels bit for bit, tools/*_bench.py); never loaded by the product
# Toolchain the kernels are written against: ROCm 7.2.0 hipcc (HIP 7.2.26015, AMD clang 22.0.0git roc-7.2.0, /opt/rocm).  Two workarounds in the sources depend on THIS
# compiler's scheduling (csrc/cmr_mfma16.h:m16_sum16 -- a DPP chain must not sink into a divergent branch; csrc/bn_linear.hip:multiply -- no
# branch between a block's last matrix instruction and the read of its accumulators); tests/test_bn_linear_gpu.py::
# test_linear_with_statistics_in_one_pass checks their symptom channel by channel -- re-run it after a compiler bump.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := $(wildcard cmr_agent_amd/csrc/*.hip)
OBJ   := $(patsubst cmr_agent_amd/csrc/%.hip,build/%.o,$(SRC))
# sources that hold an A/B switch: compiled a second time for the A/B library, every other object is shared
ABSRC := linear conv_wino conv_bf16 attention wgrad
ABOBJ := $(patsubst %,build/ab_%.o,$(ABSRC))
LIB   := cmr_agent_amd/lib/libcmr_hip.so
LIBAB := cmr_agent_amd/lib/libcmr_hip_ab.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Icmr_agent_amd/csrc

all: $(LIB) $(LIBAB)

build/%.o: cmr_agent_amd/csrc/%.hip $(wildcard cmr_agent_amd/csrc/*.h)
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

build/ab_%.o: cmr_agent_amd/csrc/%.hip $(wildcard cmr_agent_amd/csrc/*.h)
	@mkdir -p build
	$(HIPCC) $(FLAGS) -DCMR_AB_SWITCHES -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p cmr_agent_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

$(LIBAB): $(OBJ) $(ABOBJ)
	@mkdir -p cmr_agent_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(filter-out $(patsubst %,build/%.o,$(ABSRC)),$(OBJ)) $(ABOBJ)

clean:
	rm -rf build $(LIB) $(LIBAB)

.PHONY: all clean
