#!/bin/bash
# tools/ab/libscvote_timeline.so: the product objects with csrc/scvote_sort.hip rebuilt under -DSCV_SORT_TIMELINE (tools/sort_timeline.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/o1_inference_scaling_laws_amd/csrc; B=$C/build
python -c "import sys; sys.path.insert(0, '$R'); import o1_inference_scaling_laws_amd._build as b; b.build()"
mkdir -p $R/tools/ab
objs=$(ls $B/*.o | grep -v scvote_sort.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSCV_SORT_TIMELINE -c -o /tmp/scvote_sort_tl.o $C/scvote_sort.hip &
# ... and with the copy's pieces issued back to back before the sort (stamped with the loop control), to price the pieces inside the sort
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSCV_SORT_TIMELINE -DSCV_SORT_NOSPREAD -c -o /tmp/scvote_sort_tl2.o $C/scvote_sort.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libscvote_timeline.so $objs /tmp/scvote_sort_tl.o -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libscvote_timeline_nospread.so $objs /tmp/scvote_sort_tl2.o -ldl
ls -la $R/tools/ab/libscvote_timeline*.so
