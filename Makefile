# Convenience targets (the build itself lives in pygim_amd/csrc/Makefile and oracle/Makefile).
PY ?= python3

build:            ## HIP library (gfx950), TORCH_LIBRARY shims, CPU oracle, reference builds under oracle/_ref
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:             ## CPU suite (oracle, golden vectors, ABI, surface, gloo)
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## parity through the C ABI on an MI355X
	$(PY) -m pytest tests -q -m gpu

bench:            ## one JSON line: Reddit-shaped CSR SpMM, h = 256, fp32
	$(PY) bench.py

smoke:
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

.PHONY: build test test-gpu bench smoke
