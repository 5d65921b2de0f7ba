# Same targets as the reference's makefile (all q1 q2 q3 run-q1 run-q2 run-q3 run-all analyze clean
# clean-all); the toolchain is hipcc for gfx950 and the executables link the C-ABI engine library.
HIPCC ?= /opt/rocm/bin/hipcc
CXX   ?= g++
PKG   = monte-carlo-simulation-of-hull-white-model-and-sensitivities-computation_amd
LIB   = $(PKG)/libhw_hip.so
CXXFLAGS = -O2 -std=c++17 -Wall -Wextra -I include
LDFLAGS  = -pthread -L$(PKG) -lhw_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)'
SRC = src
BIN = bin
# the reference pins CUDA_VISIBLE_DEVICES=2 (makefile:27); here the device is a flag: make run-q1 ARGS="--device 2"
ARGS ?=

all: q1 q2 q3

lib: $(LIB)

$(LIB): $(PKG)/csrc/hw_engine.hip $(PKG)/csrc/hw_kernels.hpp $(PKG)/csrc/hw_model.hpp $(PKG)/csrc/hw_stream_guard.hpp $(PKG)/csrc/hw_fleet.hpp include/hw_mc.h include/hw_mc_diag.h
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -Wshadow -fno-slp-vectorize -fPIC -shared $(PKG)/csrc/hw_engine.hip -o $@

q1: $(SRC)/1_bond_pricing.cpp $(SRC)/hw_app.hpp $(LIB)
	@mkdir -p $(BIN) data plots
	$(CXX) $(CXXFLAGS) $< -o $(BIN)/$@ $(LDFLAGS)

q2: $(SRC)/2_option_pricing.cpp $(SRC)/hw_app.hpp $(LIB)
	@mkdir -p $(BIN) data plots
	$(CXX) $(CXXFLAGS) $< -o $(BIN)/$@ $(LDFLAGS)

q3: $(SRC)/3_sensitivity_analysis.cpp $(SRC)/hw_app.hpp $(LIB)
	@mkdir -p $(BIN) data plots
	$(CXX) $(CXXFLAGS) $< -o $(BIN)/$@ $(LDFLAGS)

run-q1: q1
	@mkdir -p data plots
	./$(BIN)/q1 $(ARGS)

run-q2: q2
	@mkdir -p data plots
	./$(BIN)/q2 $(ARGS)

run-q3: q3
	@mkdir -p data plots
	./$(BIN)/q3 $(ARGS)

# the reference's optional reduction-strategy benchmark (src/benchmark_reductions.cu), re-done for gfx950;
# `make microbench` builds the instruction-issue microbenchmarks under tools/
benchmark: $(SRC)/benchmark_reductions.hip $(LIB)
	@mkdir -p $(BIN) data plots
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include $< -o $(BIN)/$@ $(LDFLAGS)

# the headline metric from plain C++ over the C ABI (no Python, no torch): synchronous calls against hw_pipe_*
throughput: $(SRC)/throughput.cpp $(LIB)
	@mkdir -p $(BIN)
	$(CXX) $(CXXFLAGS) $< -o $(BIN)/$@ $(LDFLAGS)

run-throughput: throughput
	./$(BIN)/throughput $(ARGS)

run-benchmark: benchmark
	@mkdir -p data plots
	./$(BIN)/benchmark $(ARGS)

microbench:
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/bin/valu_rates
	$(HIPCC) --offload-arch=gfx950 -O3 tools/labs/valu_patterns.hip -o tools/bin/valu_patterns
	$(HIPCC) --offload-arch=gfx950 -O3 tools/labs/coissue_probe.hip -o tools/bin/coissue_probe
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I $(PKG)/csrc tools/labs/loop_lab.hip -o tools/bin/loop_lab
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I $(PKG)/csrc tools/timeline_probe.hip -o tools/bin/timeline_probe

run-all: run-q1 run-q2 run-q3

analyze: run-all run-benchmark
	@echo "data/ now holds every file the reference's analyze.py reads (same names and schemas)"

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(BIN) data/*.json data/*.csv data/summary.txt plots/*.png

clean-all:
	rm -rf $(BIN) data plots

.PHONY: all lib oracle benchmark run-benchmark microbench clean clean-all run-q1 run-q2 run-q3 run-all analyze
