cd $GRAFT_REPO_ROOT; bash tools/gpu_round.sh 2>&1 | tee gpurun_out/gpu_round_r06.log
