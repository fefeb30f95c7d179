#!/bin/bash
# build_variant.sh <name> <unit[,unit...]> <flag> [flag ...]: tools/ab/libscvote_<name>.so = the product objects with the named translation units of csrc/
# rebuilt under the extra flags (-DSCV_... experiment switches), for same-box A/B runs through SCV_LIB_PATH (tools/gpu_ab.sh, tools/regimes.py, tools/prefix_small.py).
set -e
name=$1; units=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/o1_inference_scaling_laws_amd/csrc; B=$C/build
python -c "import sys; sys.path.insert(0, '$R'); import o1_inference_scaling_laws_amd._build as b; b.build()"
mkdir -p $R/tools/ab; T=$(mktemp -d)
objs=$(ls $B/*.o)
for u in ${units//,/ }; do
  objs=$(echo "$objs" | grep -v "/$u.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c -o $T/$u.o $C/$u.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libscvote_$name.so $objs $T/*.o -ldl
rm -rf $T; ls -la $R/tools/ab/libscvote_$name.so
