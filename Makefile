# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
PKG := chroma-subsampling-image-compressor_amd

build:
	$(MAKE) -C $(PKG)/csrc
	$(MAKE) -C oracle

test: build
	python -m pytest tests -x -q -m "not gpu"

test-gpu: build
	python -m pytest tests -x -q -m gpu

bench: build
	python bench.py

asm:
	$(MAKE) -C $(PKG)/csrc asm

ubench: build
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I$(PKG)/csrc tools/ubench.hip \
		$(PKG)/csrc/csic_host.cpp $(PKG)/csrc/csic_png.cpp $(PKG)/csrc/csic_inflate.cpp -lz -pthread -o tools/ubench

# every developer micro-benchmark under tools/ (binaries are git-ignored; they travel to the GPU box with gpurun)
HIPCC_TOOL = /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I$(PKG)/csrc
tools: ubench
	for t in ubench_k ubench_overlap ubench_aql ubench_order; do \
		$(HIPCC_TOOL) tools/$$t.hip $(PKG)/csrc/csic_host.cpp $(PKG)/csrc/csic_png.cpp $(PKG)/csrc/csic_inflate.cpp -lz -pthread -L/opt/rocm/lib -lhsa-runtime64 -o tools/$$t || exit 1; done
	$(HIPCC_TOOL) tools/ubench_rows.hip -o tools/ubench_rows

clean:
	$(MAKE) -C $(PKG)/csrc clean
	$(MAKE) -C oracle clean
	rm -f tools/ubench tests/cpp/host_test

.PHONY: build test test-gpu bench asm ubench tools clean
