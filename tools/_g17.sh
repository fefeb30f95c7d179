set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== PMC sort kernel (final)"; SHAPES="3200000:4:8 1600000:4:16 800000:4:32 400000:4:64 800000:4:30" timeout 1200 bash tools/prof_regimes.sh sort3 2>&1 | grep "^## \|^\`\|fractions\|per wave\|SQ_LDS\|FETCH"
