# Convenience targets (the driver uses __graft_entry__.build / pytest / bench.py directly)
PY ?= python

.PHONY: build test-cpu test-gpu test-asan bench golden clean
build:            ## hipcc --offload-arch=gfx950 -> tightly_coupled_sfm_amd/libtcsfm_hip.so, gcc -> oracle/_build/*.so
	$(PY) -c "import __graft_entry__ as g; g.build()"
test-cpu: build   ## oracle vs the reference's golden vectors, ABI, host logic (no GPU needed)
	$(PY) -m pytest tests -q -m "not gpu"
test-gpu: build   ## HIP path vs oracle / golden vectors (needs an MI355X)
	$(PY) -m pytest tests -q -m gpu
test-asan: build ## AddressSanitizer + UBSan over the C oracle, the host SE(3) routines and examples/c_caller.c -> profiles/r05_asan.txt
	bash scripts/run_asan.sh
bench: build      ## one JSON line: frame-pairs/s + roofline + cpu_baseline
	$(PY) bench.py
golden:           ## regenerate tests/golden/*.npz by running the reference (needs /root/reference; build container only)
	$(PY) tests/golden/make_golden.py
clean:
	rm -f tightly_coupled_sfm_amd/libtcsfm_hip.so oracle/_build/*.so
