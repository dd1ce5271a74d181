#!/bin/bash
# Experiment for profiles/r03/packed_fp32_followup.md: the SLP (packed-fp32) build of mpgan_conv_mfma.hip taken through
# assembly and edited by slp_asm_edit.py, one library per spec "name:MODE[:kernel-substring]" ("none" = unedited):
#   tools/experiments/slp_asm_variants.sh asm0:none expall:expand_all "expsmall:expand_all:conv_small_kernel"
# -> tools/variants/libmpgan_<name>.so
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
cd "$HERE/../../multi-pass-gan_amd/csrc"
OUT=../../tools/variants; T=$(mktemp -d); mkdir -p $OUT
LLVM=/opt/rocm/lib/llvm/bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I../../include -I."
/opt/rocm/bin/hipcc $F --cuda-device-only -S mpgan_conv_mfma.hip -o $T/slp.s 2>/dev/null
for spec in "$@"; do
  (
  IFS=: read -r v mode only <<< "$spec"
  if [ "$mode" = none ]; then cp $T/slp.s $T/$v.s; else python3 $HERE/slp_asm_edit.py $T/slp.s $T/$v.s $mode $only; fi
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/$v.s -o $T/$v.dev.o
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/$v.out $T/$v.dev.o
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input=$T/$v.out -output=$T/$v.hipfb
  /opt/rocm/bin/hipcc $F --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/$v.hipfb -c mpgan_conv_mfma.hip -o $T/$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC mpgan_api.o $T/$v.o mpgan_elem.o mpgan_train.o mpgan_wgrad_mfma.o mpgan_tiles.o -o $OUT/libmpgan_$v.so
  ) &
done
wait
rm -rf $T
ls -la $OUT/
