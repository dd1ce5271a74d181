#!/bin/bash
# builds tools/variants/libmpgan_<name>.so: the product library with other settings of the K-loop switches
set -e
cd "$(dirname "$0")/../multi-pass-gan_amd/csrc"
OUT=../../tools/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -fno-vectorize -I../../include -I."
build() {  # name, extra flags
  name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c mpgan_conv_mfma.hip -o $OUT/conv_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC mpgan_api.o $OUT/conv_$name.o mpgan_elem.o mpgan_train.o mpgan_wgrad_mfma.o mpgan_tiles.o -o $OUT/libmpgan_$name.so
  rm $OUT/conv_$name.o
}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  build $name $flags &
done
wait
ls -la $OUT
