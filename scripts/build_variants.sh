#!/bin/bash
# Build A/B variants of libctamd.so that differ in ct_unet.hip compile-time switches (for one-call comparisons on the GPU box):
#   usage: scripts/build_variants.sh name1 "-DFOO=1 -DBAR=2" name2 "-DFOO=0" ...   ->  3deecelltracker_amd/_variants/libctamd_<name>.so
# run with CTAMD_LIB=3deecelltracker_amd/_variants/libctamd_<name>.so
set -eu
cd "$(dirname "$0")/../3deecelltracker_amd/csrc"
make -s all
mkdir -p ../_variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall $flags -c ct_unet.hip -o ../_variants/ct_unet_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_variants/libctamd_$name.so ct_common.o ../_variants/ct_unet_$name.o ct_match.o ct_preprocess.o ct_correct.o ct_segment.o &&
    echo "built $name ($flags)" ) &
done
wait
