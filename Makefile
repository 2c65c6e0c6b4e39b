# Builds the HIP libraries (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
#   cmr_agent_amd/lib/libcmr_hip.so      the product: include/cmr_hip.h, no mutable state
#   cmr_agent_amd/lib/libcmr_hip_ab.so   the same sources with -DCMR_AB_SWITCHES: + the kernel-variant switches of include/cmr_hip_ab.h
#                                        (tests that compare two kernels bit for bit, tools/*_bench.py); never loaded by the product
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
