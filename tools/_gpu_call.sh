#!/bin/bash
# scratch entry point of a gpurun call (gpurun --timeout N -- 'bash tools/_gpu_call.sh'): edited per call
cd /root/repo
for lib in tools/ab/libscvote_base.so tools/ab/libscvote_base.so ""; do
  echo "== test_first_call_of_a_fresh_context_is_bit_exact, 240 seeds; library: ${lib:-final}"
  t0=$SECONDS; SCV_LIB_PATH=$lib timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "fresh_context" 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -12; echo "   $((SECONDS-t0)) s"
done
echo "== ... 4000 further seeds on the final library"
t0=$SECONDS; SCV_FUZZ_FIRST=1000 SCV_FUZZ_FRESH_SEEDS=4000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "fresh_context" 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -12; echo "   $((SECONDS-t0)) s"
