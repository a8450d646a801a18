# Builds libcognn_hip.so (the HIP engine + C ABI) for gfx950, in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := cognn_amd/csrc
HOST := cognn_amd/host
OUT := cognn_amd/libcognn_hip.so
HOSTCXX ?= g++
HOSTFLAGS := -O2 -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude -I$(CSRC)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude -I$(CSRC)
# make ABLATION=1: also build the timing-only (wrong-result) variants of the Beaver GEMM selected by COGNN_GEMM_DBG
ifdef ABLATION
HIPFLAGS += -DCOGNN_GEMM_ABLATION
endif
KERNEL_SRCS := $(CSRC)/capi.hip $(CSRC)/kernels_elementwise.hip $(CSRC)/kernels_gather.hip $(CSRC)/kernels_gemm.hip $(CSRC)/exchange_rccl.hip $(CSRC)/graph_build.hip
HOST_SRCS := $(filter-out $(HOST)/harness_main.cpp,$(wildcard $(HOST)/*.cpp))
HARNESS := bin/gcn-optimize
OBJS := $(KERNEL_SRCS:.hip=.o) $(HOST_SRCS:.cpp=.o)

all: $(OUT) $(HARNESS)

# the reference's command-line entry point (harness.cpp) over the engine; gcn-inference-optimize and gcn-original are the same binary
$(HARNESS): $(HOST)/harness_main.cpp $(OUT) $(HOST)/graph.h include/cognn_engine.h include/cognn_exchange.h
	mkdir -p bin
	$(HOSTCXX) $(HOSTFLAGS) -o $@ $(HOST)/harness_main.cpp -Lcognn_amd -lcognn_hip -Wl,-rpath,'$$ORIGIN/../cognn_amd' -Wl,-rpath,/opt/rocm/lib
	ln -sf gcn-optimize bin/gcn-inference-optimize
	ln -sf gcn-optimize bin/gcn-original

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/pair_chain.h $(CSRC)/cognn_spec.h include/cognn_hip.h include/cognn_engine.h include/cognn_exchange.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(HOST)/%.o: $(HOST)/%.cpp $(wildcard $(HOST)/*.h) $(CSRC)/cognn_spec.h include/cognn_hip.h include/cognn_engine.h
	$(HOSTCXX) $(HOSTFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -L/opt/rocm/lib -lrccl -lpthread

clean:
	rm -f $(OBJS) $(OUT) bin/gcn-optimize bin/gcn-inference-optimize
.PHONY: all clean
