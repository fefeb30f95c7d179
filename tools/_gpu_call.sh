set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=5 --tb=short -k "short_cells" 2>&1 | tail -15
echo "== ... 6000 further seeds of test_random_short_cells_in_device_memory_both_record_forms (DEVICE memory, N <= 127, 16-byte / 4-byte records / counters only)" | tee -a gpurun_out/fuzz_extended.log
SCV_FUZZ_FIRST=300 SCV_FUZZ_CELL_SEEDS=6000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=5 --tb=short -k "short_cells" 2>&1 | tail -4 | tee -a gpurun_out/fuzz_extended.log
