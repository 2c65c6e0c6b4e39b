# Builds the HIP library (gfx950 only) in-tree.  hipcc cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := $(wildcard cmr_agent_amd/csrc/*.hip)
OBJ   := $(patsubst cmr_agent_amd/csrc/%.hip,build/%.o,$(SRC))
LIB   := cmr_agent_amd/lib/libcmr_hip.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Icmr_agent_amd/csrc

all: $(LIB)

build/%.o: cmr_agent_amd/csrc/%.hip cmr_agent_amd/csrc/cmr_common.h cmr_agent_amd/csrc/cmr_chain.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p cmr_agent_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

clean:
	rm -rf build $(LIB)

.PHONY: all clean
