#!/bin/bash
cd /root/repo
echo "== parity (prefix, sort, token, lane, fuzz)"
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "prefix or token or sort or lane or random" 2>&1 | tail -3
for lib in tools/ab/libscvote_head.so ""; do
  export SCV_LIB_PATH=$lib
  echo "== library: ${lib:-new}: general kernels alone (prefix_path = 1: one lane per problem up to 64 votes)"
  timeout 600 python tools/prefix_small.py prefix_path=1 2>&1 | grep -v amdgpu.ids | head -8 | cut -c1-140
  echo "== tokens 128 promised"
  for P in 50000 200000; do python tools/one_case.py --prefix --tokens --rounds 11 --P $P --N 128 --opt prefix_path=5 2>&1 | grep -v amdgpu.ids | cut -c1-140; done
done
