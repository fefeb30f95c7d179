#!/bin/bash
# build_ab.sh <git-rev> <name>: the library of an earlier commit as tools/ab/libscvote_<name>.so (git-ignored, travels to the GPU box), for
# same-box A/B runs: SCV_LIB_PATH=tools/ab/libscvote_<name>.so python tools/regimes.py ...
set -e
rev=$1; name=$2
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$R" archive "$rev" o1_inference_scaling_laws_amd/csrc include | tar -x -C "$T"
C=$T/o1_inference_scaling_laws_amd/csrc
objs=()
for u in "$C"/*.hip; do
  o=${u%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -o "$o" "$u" &
  objs+=("$o")
done
wait
mkdir -p "$R/tools/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/ab/libscvote_$name.so" "${objs[@]}" -ldl
rm -rf "$T"
ls -la "$R/tools/ab/libscvote_$name.so"
