cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; timeout 1500 bash tools/tsan_host.sh 2>&1 | tee gpurun_out/tsan_host.log | tail -60
