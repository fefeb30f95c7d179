# Convenience targets (no build system dependency: hipcc + gcc + python only).
.PHONY: build test-cpu test-gpu bench round golden clean

build:            ## hipcc --offload-arch=gfx950 -> csrc/libscvote.so ; gcc -> oracle/libscv_oracle.so
	python -c "import __graft_entry__ as g; g.build()"

test-cpu: build   ## oracle vs goldens, host logic, ABI symbols, gloo world sizes 2-3 (no GPU needed)
	python -m pytest tests -x -q -m "not gpu"

test-gpu:         ## parity through the C ABI on an MI355X
	python -m pytest tests -x -q -m gpu

bench:            ## one JSON line
	python bench.py

round:            ## on a GPU box: tests, smoke, bench, regimes, C5, rocprofv3 + PMC passes -> gpurun_out/
	bash tools/gpu_round.sh

golden:           ## regenerate tests/golden/ from the UNMODIFIED reference (needs /root/reference)
	python tests/golden/make_golden.py

clean:
	rm -f o1_inference_scaling_laws_amd/csrc/libscvote.so oracle/libscv_oracle.so tools/hbm_probe.bin
