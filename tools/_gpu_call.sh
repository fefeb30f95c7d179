#!/bin/bash
# scratch entry point of a gpurun call (gpurun --timeout N -- 'bash tools/_gpu_call.sh'): edited per call
cd /root/repo
echo "== extended fuzz on the FINAL library (24- / 56-vote shapes, split-scratch fix), the fuzz file alone in a fresh process (HOST-mode calls on the context's own stream): seeds 100000 .. 119999 of test_random_configuration_is_bit_exact, 100000 .. 107999 of the prefix fuzz, 100000 .. 102999 of the DEVICE-memory cells"
t0=$SECONDS; SCV_FUZZ_FIRST=100000 SCV_FUZZ_SEEDS=20000 SCV_FUZZ_PREFIX_SEEDS=8000 SCV_FUZZ_CELL_SEEDS=3000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -6; echo "   $((SECONDS-t0)) s"
echo "== the full GPU suite in REVERSE order (SCV_TEST_ORDER=reverse)"
t0=$SECONDS; SCV_TEST_ORDER=reverse timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -8; echo "   $((SECONDS-t0)) s"
echo "== PMC of the 24- and 56-vote sorted shapes"
SHAPES="1000000:4:24 400000:4:56" timeout 600 bash tools/prof_regimes.sh r06_sort24_56 2>&1 | grep -v amdgpu.ids | tail -40
