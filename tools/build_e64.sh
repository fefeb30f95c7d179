#!/bin/bash
# build_e64.sh <out.so>: the library with every `v_cndmask_b32_e32 d, a, b, vcc` of the device code re-encoded as VOP3 (`v_cndmask_b32_e64`).
# Why: on gfx950 the VOP2 form issues every ~16 cycles per SIMD, the VOP3 form every ~4.6 (tools/valu_probe.hip, profiles/r04_valu_probe.log);
# hipcc's shrink pass picks the VOP2 form whenever the mask is in vcc.  Same instruction, same operands, 4 bytes longer.
set -e
out=$1
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/o1_inference_scaling_laws_amd/csrc
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
one() {
  u=$1; b=$(basename $u .hip)
  /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o $T/$b.s $u
  sed -E 's/\bv_cndmask_b32_e32 (.*), vcc$/v_cndmask_b32_e64 \1, vcc/' $T/$b.s > $T/$b.e64.s
  $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/$b.e64.s -o $T/$b.devobj
  $L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/$b.co $T/$b.devobj
  $L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/$b.co -output=$T/$b.hipfb
  /opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/$b.hipfb -c $u -o $T/$b.o
}
for u in $C/*.hip; do one $u & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $T/*.o -ldl
grep -c "v_cndmask_b32_e64 .*, vcc$" $T/*.e64.s | tr '\n' ' '; echo
rm -rf $T
ls -la $out
