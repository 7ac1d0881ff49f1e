#!/bin/bash
# Developer tool ("price the s_nop padding"): builds a copy of libbjj_hip.so in which the `s_nop 0` that
# hipcc places after every inline-asm statement whose result the next instruction reads is REMOVED from the device assembly.
# hipcc pads because it must assume the asm could hold an SDWA / op_sel instruction (dst_sel forwarding hazard of gfx940+);
# the statements here hold only v_mad_u64_u32, which has no such hazard, so the stripped code is still correct -- it is a
# TIMING build for the A/B, not a product.   usage: tools/build_stripped_snop.sh <out.so> [extra hipcc flags]
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
CSRC=$ROOT/babyjubjub-rs_amd/csrc
OUT=$(realpath -m $1); shift
EXTRA="$@"
LLVM=/opt/rocm/lib/llvm/bin
B=$CSRC/build_strip; mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -Wno-unused-value $EXTRA"
cd $CSRC
build_unit() {
  u=$1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -S --cuda-device-only -o $B/$u.s $u.hip 2>/dev/null
  python3 - $B/$u.s <<'PY'
import sys
p = sys.argv[1]
L = open(p).read().split("\n")
out, n = [], 0
for i, l in enumerate(L):
    if l.strip() == "s_nop 0" and i > 0 and L[i - 1].strip() == ";;#ASMEND":
        n += 1
        continue
    out.append(l)
open(p, "w").write("\n".join(out))
print("%s: removed %d s_nop" % (p, n))
PY
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $B/$u.s -o $B/$u.dev.o
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B/$u.out $B/$u.dev.o
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input=$B/$u.out -output=$B/$u.hipfb
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $B/$u.hipfb -c -o $B/$u.o $u.hip 2>/dev/null
}
for u in k_fixed k_var k_hash_codec k_verify k_sign; do build_unit $u & done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c -o $B/bjj_hip.o bjj_hip.hip 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $B/k_fixed.o $B/k_var.o $B/k_hash_codec.o $B/k_verify.o $B/k_sign.o $B/bjj_hip.o -ldl -lpthread
ls -la $OUT
