#!/bin/bash
# BEST EFFORT on a GPU box: the HOST-mode tests through csrc/libscvote_tsan.so (host code of scvote.hip / scvote_comm.hip under
# -fsanitize=thread, device code unchanged; built here by `python -m o1_inference_scaling_laws_amd._build tsan`).  The HIP runtime and
# torch are not TSAN-instrumented and ROCm reserves address ranges TSAN's shadow wants, so this may not start at all -- the log says
# which.  The binding check of the library's threads is the CPU harness (tests/test_host_sanitizers.py), which always runs.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
export SCV_LIB_PATH=$R/o1_inference_scaling_laws_amd/csrc/libscvote_tsan.so
export TSAN_OPTIONS="report_signal_unsafe=0 ignore_noninstrumented_modules=1 history_size=4 exitcode=0 log_path=$R/gpurun_out/tsan_report"
echo "runtime: $RT"; echo "library: $SCV_LIB_PATH"
LD_PRELOAD=$RT timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
  -k "test_host_mode_pipeline_and_pinned_sources or test_host_mode_streams_problem_chunks or test_no_cpp_exception_crosses_the_abi" 2>&1 | tail -15
echo "exit: $?"
ls gpurun_out/tsan_report* 2>/dev/null | head; for f in gpurun_out/tsan_report*; do [ -f "$f" ] && grep -c "WARNING: ThreadSanitizer" $f; done
