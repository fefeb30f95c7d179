set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== extended fuzz on the final library (round 6): seeds 800 .. 20799 of test_random_configuration_is_bit_exact, 800 .. 5799 of the prefix fuzz" | tee gpurun_out/fuzz_extended.log
SCV_FUZZ_FIRST=800 SCV_FUZZ_SEEDS=20000 SCV_FUZZ_PREFIX_SEEDS=5000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=5 --tb=short 2>&1 | tail -4 | tee -a gpurun_out/fuzz_extended.log
