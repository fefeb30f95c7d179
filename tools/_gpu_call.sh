#!/bin/bash
cd /root/repo
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== parity (prefix, sort, short, fuzz)"
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "prefix or sort or short or random or token" 2>&1 | tail -4
