#!/bin/bash
# The HOST-mode host code of the library under ThreadSanitizer ON A GPU BOX: csrc/libscvote_tsan.so (scvote.hip / scvote_comm.hip with
# -Xarch_host -fsanitize=thread, device code unchanged; `python -m o1_inference_scaling_laws_amd._build tsan`) driven by tools/tsan_host_driver.py
# (ctypes + numpy, no torch: torch does not import under a preloaded TSAN runtime).  The HIP runtime is not instrumented
# (ignore_noninstrumented_modules=1); reports about the library's own frames are what counts.  One run per fault mode, each with its own report
# file; the last mode, `race`, plants a DELIBERATE unsynchronised counter among the copy pieces: its report proves the detector is live in this
# setup -- the other modes must produce no report at all.  Besides this, the threads of the library are checked on the CPU by
# tests/test_host_sanitizers.py (gcc -fsanitize=thread / address over csrc/scvote_hostpool.h), which always runs.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
export SCV_LIB_PATH=$R/o1_inference_scaling_laws_amd/csrc/libscvote_tsan.so
rm -f $R/gpurun_out/tsan_report*
echo "runtime: $RT"; echo "library: $SCV_LIB_PATH"
for fault in none thread alloc throw race; do
  echo "== SCV_TEST_FAULT=$fault"
  f=$fault; [ $f = none ] && f=""
  TSAN_OPTIONS="report_signal_unsafe=0 ignore_noninstrumented_modules=1 history_size=4 exitcode=0 log_path=$R/gpurun_out/tsan_report_$fault" \
    SCV_TEST_FAULT=$f LD_PRELOAD=$RT timeout 600 python tools/tsan_host_driver.py 2>&1 | tail -14
done
echo "== ThreadSanitizer reports per run"
status=0
for fault in none thread alloc throw race; do
  c=0; for f in gpurun_out/tsan_report_$fault.*; do [ -f "$f" ] && c=$((c + $(grep -c "WARNING: ThreadSanitizer" $f))); done
  echo "SCV_TEST_FAULT=$fault: $c warning(s)"
  if [ $fault = race ]; then
    [ $c -gt 0 ] || { echo "  the deliberate race was NOT reported: the detector is not live"; status=1; }
    other=$(cat gpurun_out/tsan_report_race.* 2>/dev/null | grep -A2 "WARNING: ThreadSanitizer" | grep "#0 " | grep -vc "host_pipelined")
    echo "  reports whose first frame is not the planted lambda of host_pipelined: $other"
  else
    [ $c -eq 0 ] || { status=1; for f in gpurun_out/tsan_report_$fault.*; do grep -A12 "WARNING: ThreadSanitizer" $f | head -40; done; }
  fi
done
[ $status -eq 0 ] && echo "TSAN VERDICT: clean (no report from the library's own host code; the planted race is seen)" || echo "TSAN VERDICT: see the reports above"
