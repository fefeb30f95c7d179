set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
SHAPES="200000:4:128 100000:4:256 50000:4:1024" DIST=1 timeout 900 bash tools/prof_regimes.sh reg_r3_d1 2>&1 | grep "^## \|^\`\|fractions\|per wave\|SQ_LDS\|FETCH"
SHAPES="100000:4:256 50000:4:1024" DIST=3 timeout 900 bash tools/prof_regimes.sh reg_r3_d3 2>&1 | grep "^## \|^\`\|fractions\|per wave\|SQ_LDS"
