#!/bin/bash
# scratch entry point of a gpurun call (gpurun --timeout N -- 'bash tools/_gpu_call.sh'): edited per call
cd /root/repo
echo "== sorted cells (every shape x 6 distributions), packed records, every fuzz test on the final library (whole waves per SIMD in scv_sort_cells)"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -k "sorted_cells or fuzz or packed_cell or fresh_context or graph" 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-400
