#!/bin/bash
# per-layer U-Net times of A/B builds of libctamd.so (scripts/build_variants.sh), interleaved twice:  ab_layers.sh name1 name2 ...   ("shipped" = the in-tree library)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = shipped ]; then unset CTAMD_LIB; else export CTAMD_LIB=$PWD/3deecelltracker_amd/_variants/libctamd_$v.so; fi
    echo "== $v (pass $rep)"
    python scripts/microbench.py unet --layers 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | sed 's/ms\/vol (1 launches)//g; s/  */ /g'
    echo
  done
done
