#!/bin/bash
# builds tools/variants/libmpgan_<name>.so: the product library with other settings of the K-loop switches
set -e
cd "$(dirname "$0")/../multi-pass-gan_amd/csrc"
OUT=../../tools/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -fno-vectorize -I../../include -I."
build() {  # name, extra flags ("@slp" among them: build WITH the SLP / loop vectorisers, i.e. packed fp32 allowed)
  name=$1; shift
  local F="$FLAGS" args=()
  for a in "$@"; do if [ "$a" = "@slp" ]; then F="${FLAGS/-fno-slp-vectorize -fno-vectorize/}"; else args+=("$a"); fi; done
  set -- "${args[@]}"
  FLAGS_USED="$F"
  /opt/rocm/bin/hipcc $F "$@" -c mpgan_conv_mfma.hip -o $OUT/conv_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC mpgan_api.o $OUT/conv_$name.o mpgan_elem.o mpgan_train.o mpgan_wgrad_mfma.o mpgan_tiles.o -o $OUT/libmpgan_$name.so
  rm $OUT/conv_$name.o
}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  build $name $flags &
done
wait
ls -la $OUT
